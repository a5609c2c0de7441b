"""Wall boxes from ScanNet-Planes quads: same-name mirror of reference gapro/scannet_planes.py.

Host NumPy, as in the reference (this is per-scene preparation of a handful of quads, not a hot
loop).  ``get_wall_boxes(scan_name)`` keeps the reference signature and its relative dataset paths
(scannet_planes.py:163,178); ``data_root`` is an additive option.
"""
from __future__ import annotations

import json
import os

import numpy as np


def isFourPointsInSamePlane(p0, p1, p2, p3, error):
    """scannet_planes.py:8-22: |triple product| <= error."""
    s1, s2, s3 = p1 - p0, p2 - p0, p3 - p0
    result = float(np.dot(s1, np.cross(s2, s3)))
    return bool(result - error <= 0 <= result + error)


def get_normal(quad_vert, center=None):
    """scannet_planes.py:25-55: least-squares plane z = a x + b y + c, or a vertical fit a x + b y = -1."""
    quad_vert = np.asarray(quad_vert, dtype=np.float64)
    A = np.c_[quad_vert[:4, 0], quad_vert[:4, 1], np.ones(4)]
    b = quad_vert[:4, 2]
    temp = A.T @ A
    if np.linalg.det(temp) > 1e-10:
        fit = np.linalg.inv(temp) @ A.T @ b
        normal = np.array([fit[0] / fit[2], fit[1] / fit[2], -1.0 / fit[2]])
    else:
        A2 = A[:, 0:2]
        fit = np.linalg.inv(A2.T @ A2) @ A2.T @ np.array([-1.0, -1.0, -1.0, -1.0])
        normal = np.array([fit[0], fit[1], 0.0])
    return normal / np.linalg.norm(normal)


def get_center(verts):
    return np.mean(np.array(verts), axis=0)


def get_box_from_quad(quad_vert, center=None):
    """scannet_planes.py:101-159: AABB of a vertical quad = centre +- width/2 along the in-plane
    horizontal direction, +- height/2 along z."""
    quad_vert = np.asarray(quad_vert, dtype=np.float64)
    quad_center = np.mean(quad_vert, axis=0)
    n = get_normal(quad_vert, center)
    v = np.array([n[0], n[1], 0.0])
    v = v / np.linalg.norm(v)
    edge = quad_vert[0] - quad_vert[1]
    # torch.cosine_similarity(edge, [0,0,1]) with its eps=1e-8 clamp on the norms
    cos_theta = edge[2] / max(np.linalg.norm(edge), 1e-8)
    l1 = np.linalg.norm(quad_vert[0] - quad_vert[1])
    l2 = np.linalg.norm(quad_vert[1] - quad_vert[2])
    l3 = np.linalg.norm(quad_vert[2] - quad_vert[3])
    l4 = np.linalg.norm(quad_vert[3] - quad_vert[0])
    l5, l6 = (l1 + l3) / 2, (l2 + l4) / 2
    height, width = (l5, l6) if abs(cos_theta) > 0.5 else (l6, l5)
    v = v / max(np.linalg.norm(v), 1e-6)
    x1 = quad_center[0] + width * v[1] / 2
    x2 = quad_center[0] - width * v[1] / 2
    y1 = quad_center[1] - width * v[0] / 2
    y2 = quad_center[1] + width * v[0] / 2
    h1 = quad_center[2] + height / 2
    h2 = quad_center[2] - height / 2
    return np.array([min(x1, x2), min(y1, y2), min(h1, h2), max(x1, x2), max(y1, y2), max(h1, h2)])


def read_axis_align_matrix(meta_file):
    """The 'axisAlignment = ...' line parser shared by gen_ps.py:58-64 and scannet_planes.py:162-169."""
    axis_align_matrix = None
    for line in open(meta_file).readlines():
        if "axisAlignment" in line:
            axis_align_matrix = [float(x) for x in line.rstrip().strip("axisAlignment = ").split(" ")]
            break
    if axis_align_matrix is None:
        raise ValueError("no axisAlignment line in " + meta_file)
    return np.array(axis_align_matrix).reshape((4, 4))


def transform(scan_name, mesh_vertices, data_root="dataset/scannetv2"):
    """scannet_planes.py:162-174."""
    A = read_axis_align_matrix(os.path.join(data_root, "scans_transform", scan_name, scan_name + ".txt"))
    pts = np.ones((mesh_vertices.shape[0], 4))
    pts[:, 0:3] = mesh_vertices[:, 0:3]
    pts = np.dot(pts, A.transpose())
    mesh_vertices[:, 0:3] = pts[:, 0:3]
    return mesh_vertices


def get_wall_boxes(scan_name, data_root="dataset/scannetv2"):
    """scannet_planes.py:177-230 -> (cls i64[W] == 18, boxes f64[W,6], volumes f64[W]) or ([], [], [])."""
    quad_file_path = os.path.join(data_root, "scannet_planes", scan_name + ".json")
    if not os.path.exists(quad_file_path):
        return [], [], []
    with open(quad_file_path, "r") as quad_file:
        plane_dict = json.load(quad_file)
    quad_dict = plane_dict["quads"]
    vert_dict = plane_dict["verts"]
    for i in range(len(vert_dict)):  # y <- -z, z <- y   (:190-193)
        temp = vert_dict[i][1]
        vert_dict[i][1] = -vert_dict[i][2]
        vert_dict[i][2] = temp
    verts = transform(scan_name, np.array(vert_dict, dtype=np.float64), data_root)
    quads = [q for q in quad_dict if len(q) == 4]
    if len(quads) == 0:
        return [], [], []
    quad_verts = np.asarray([[verts[j] for j in q] for q in quads])
    quad_verts = [qv for qv in quad_verts if isFourPointsInSamePlane(qv[0], qv[1], qv[2], qv[3], 100)]
    room_center = get_center(vert_dict)
    vertical = [qv for qv in quad_verts if abs(get_normal(qv, room_center)[2]) < 0.2]  # :213-215
    if len(vertical) == 0:
        return [], [], []
    boxes = np.array([get_box_from_quad(qv, room_center) for qv in vertical])
    cls = np.ones(boxes.shape[0]).astype(np.int64) * 18  # the reference's np.int
    volumes = np.prod(np.clip(boxes[:, 3:] - boxes[:, :3], a_min=0.0, a_max=None), axis=-1)
    return cls, boxes, volumes
