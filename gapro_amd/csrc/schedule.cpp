// Host side of the pseudo-label path: static pair schedule + merge/fallback/labels.
//
// Exact replay of the control flow of reference gapro/gen_ps_utils.py:365-476.  The reference
// drives this loop from Python with a device->host sync per `len(nonzero(...))`; which pairs get a
// GP fit and on which superpoints depends only on (boxes, bb_occupancy_spp) -- never on GP outputs
//  -- so it is enumerated here once, all fits are launched as one batch,
// and the merge is replayed afterwards in reference order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/gapro_hip.h"

namespace {

constexpr int kMaxNum = 1000000;       // gen_ps_utils.py:308
constexpr double kIouOverlap = 0.0001;  // :393
constexpr double kIouSkip = 0.6;        // :425
constexpr double kContainOffset = 0.1;  // :411,:418

struct Event {
  uint8_t kind;  // 0 contain, 1 fit
  int32_t b1, b2;
  int32_t aux;  // contain: winner box; fit: fit id
  std::vector<int32_t> inter;
};

struct Fit {
  int32_t b1, b2;
  int32_t event;
};

}  // namespace

struct gapro_schedule {
  int32_t n_spps = 0, n_boxes = 0, words = 0;
  std::vector<uint64_t> occ_bits;
  std::vector<int32_t> n_bbs;
  std::vector<std::vector<int32_t>> single;  // per box: superpoints whose only box it is (ascending)
  std::vector<Event> events;
  std::vector<Fit> fits;
};

namespace {

// IoU half of batch_giou_cross (gen_ps_utils.py:33-49), float64.
double box_iou(const double* a, const double* b) {
  double inter = 1.0, va = 1.0, vb = 1.0;
  for (int k = 0; k < 3; ++k) {
    inter *= std::max(std::min(a[3 + k], b[3 + k]) - std::max(a[k], b[k]), 0.0);
    va *= std::max(a[3 + k] - a[k], 0.0);
    vb *= std::max(b[3 + k] - b[k], 0.0);
  }
  const double uni = va + vb - inter;
  return inter / (uni + 1e-6);
}

// is_box1_in_box2 (gen_ps_utils.py:75-76) with offset 0.1.
bool box1_in_box2(const double* b1, const double* b2) {
  for (int k = 0; k < 3; ++k) {
    if (!((b1[k] + kContainOffset) >= b2[k])) return false;
    if (!((b1[3 + k] - kContainOffset) <= b2[3 + k])) return false;
  }
  return true;
}

inline uint64_t pair_key(int a, int b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; }

}  // namespace

extern "C" {

int gapro_schedule_build(int32_t n_spps, int32_t n_boxes, const double* h_boxes, const uint64_t* h_occ_bits,
                         const int32_t* h_n_bbs, gapro_schedule** out) {
  if (!out) return GAPRO_ERR_BAD_ARG;
  *out = nullptr;
  if (n_spps <= 0 || n_boxes <= 0 || !h_boxes || !h_occ_bits || !h_n_bbs) return GAPRO_ERR_BAD_ARG;
  gapro_schedule* s = new (std::nothrow) gapro_schedule();
  if (!s) return GAPRO_ERR_OOM;
  const int B = n_boxes, W = (B + 63) / 64;
  s->n_spps = n_spps;
  s->n_boxes = B;
  s->words = W;
  s->occ_bits.assign(h_occ_bits, h_occ_bits + (size_t)n_spps * W);
  s->n_bbs.assign(h_n_bbs, h_n_bbs + n_spps);
  s->single.resize(B);

  // One pass over the superpoints: single-box lists and, for multi-box superpoints, the list of
  // superpoints per (lo, hi) box pair -- both ascending in superpoint index like torch.nonzero.
  std::unordered_map<uint64_t, std::vector<int32_t>> pair_spps;
  std::vector<int32_t> set_boxes;
  for (int32_t sp = 0; sp < n_spps; ++sp) {
    const int nb = s->n_bbs[sp];
    if (nb == 0) continue;
    set_boxes.clear();
    for (int w = 0; w < W; ++w) {
      uint64_t bits = s->occ_bits[(size_t)sp * W + w];
      while (bits) {
        const int b = __builtin_ctzll(bits);
        set_boxes.push_back(w * 64 + b);
        bits &= bits - 1;
      }
    }
    if (nb == 1) {
      s->single[set_boxes[0]].push_back(sp);
    } else {
      for (size_t i = 0; i < set_boxes.size(); ++i)
        for (size_t j = i + 1; j < set_boxes.size(); ++j)
          pair_spps[pair_key(set_boxes[i], set_boxes[j])].push_back(sp);
    }
  }

  std::vector<double> iou((size_t)B * B);
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < B; ++j) iou[(size_t)i * B + j] = (i == j) ? 0.0 : box_iou(h_boxes + 6 * i, h_boxes + 6 * j);  // :385-386

  std::vector<uint8_t> visited(B, 0);  // :388
  std::vector<int32_t> cand;
  for (int b1 = 0; b1 < B; ++b1) {  // :390
    cand.clear();
    for (int b2 = 0; b2 < B; ++b2)
      if (iou[(size_t)b1 * B + b2] > kIouOverlap && !visited[b2]) cand.push_back(b2);  // :393-394 (snapshot)
    if (cand.empty()) {  // :397-399
      visited[b1] = 1;
      continue;
    }
    for (int b2 : cand) {  // :401
      auto it = pair_spps.find(pair_key(std::min(b1, b2), std::max(b1, b2)));  // :403-405
      if (it == pair_spps.end() || it->second.empty()) continue;               // :408-409
      const double* bx1 = h_boxes + 6 * b1;
      const double* bx2 = h_boxes + 6 * b2;
      if (box1_in_box2(bx1, bx2)) {  // :411-416
        s->events.push_back(Event{0, b1, b2, b1, it->second});
        visited[b1] = 1;
        break;
      }
      if (box1_in_box2(bx2, bx1)) {  // :418-423
        s->events.push_back(Event{0, b1, b2, b2, it->second});
        visited[b2] = 1;
        continue;
      }
      if (iou[(size_t)b1 * B + b2] >= kIouSkip) continue;               // :425-426
      if (s->single[b1].empty() || s->single[b2].empty()) continue;     // :428-432
      s->events.push_back(Event{1, b1, b2, (int32_t)s->fits.size(), it->second});
      s->fits.push_back(Fit{b1, b2, (int32_t)s->events.size() - 1});
    }
    visited[b1] = 1;  // :448
  }
  *out = s;
  return GAPRO_OK;
}

void gapro_schedule_free(gapro_schedule* s) { delete s; }

int gapro_schedule_get_counts(const gapro_schedule* s, gapro_schedule_counts* out) {
  if (!s || !out) return GAPRO_ERR_BAD_ARG;
  std::memset(out, 0, sizeof(*out));
  out->n_events = (int32_t)s->events.size();
  out->n_fits = (int32_t)s->fits.size();
  for (const Event& e : s->events) out->n_event_idx += (int64_t)e.inter.size();
  for (const Fit& f : s->fits) {
    const int m = (int)(s->single[f.b1].size() + s->single[f.b2].size());
    const int t = (int)s->events[f.event].inter.size();
    out->n_fit_idx += m + t;
    out->n_fit_out += t;
    out->max_m = std::max(out->max_m, m);
    out->max_t = std::max(out->max_t, t);
  }
  return GAPRO_OK;
}

int gapro_schedule_export_fits(const gapro_schedule* s, int32_t feats_row_base, int64_t idx_base, int64_t out_base,
                               int32_t scene, gapro_fit_desc* h_descs, int32_t* h_idx) {
  if (!s) return GAPRO_ERR_BAD_ARG;
  if (s->fits.empty()) return GAPRO_OK;
  if (!h_descs || !h_idx) return GAPRO_ERR_BAD_ARG;
  int64_t io = 0, oo = 0;
  for (size_t i = 0; i < s->fits.size(); ++i) {
    const Fit& f = s->fits[i];
    const std::vector<int32_t>& t1 = s->single[f.b1];
    const std::vector<int32_t>& t2 = s->single[f.b2];
    const std::vector<int32_t>& in = s->events[f.event].inter;
    gapro_fit_desc& d = h_descs[i];
    d.m1 = (int32_t)t1.size();
    d.m2 = (int32_t)t2.size();
    d.t = (int32_t)in.size();
    d.b1 = f.b1;
    d.b2 = f.b2;
    d.scene = scene;
    d.idx_offset = idx_base + io;
    d.out_offset = out_base + oo;
    d.ws_offset = 0;
    for (int32_t v : t1) h_idx[io++] = v + feats_row_base;
    for (int32_t v : t2) h_idx[io++] = v + feats_row_base;
    for (int32_t v : in) h_idx[io++] = v + feats_row_base;
    oo += d.t;
  }
  return GAPRO_OK;
}

int gapro_schedule_export_events(const gapro_schedule* s, uint8_t* h_kind, int32_t* h_b1, int32_t* h_b2,
                                 int32_t* h_aux, int64_t* h_offsets, int32_t* h_event_idx) {
  if (!s || !h_offsets) return GAPRO_ERR_BAD_ARG;
  int64_t o = 0;
  for (size_t i = 0; i < s->events.size(); ++i) {
    const Event& e = s->events[i];
    if (h_kind) h_kind[i] = e.kind;
    if (h_b1) h_b1[i] = e.b1;
    if (h_b2) h_b2[i] = e.b2;
    if (h_aux) h_aux[i] = e.aux;
    h_offsets[i] = o;
    if (h_event_idx) std::copy(e.inter.begin(), e.inter.end(), h_event_idx + o);
    o += (int64_t)e.inter.size();
  }
  h_offsets[s->events.size()] = o;
  return GAPRO_OK;
}

int gapro_schedule_merge(const gapro_schedule* s, const float* h_probs_new, const uint8_t* h_labels,
                         const float* h_mu, const float* h_var, const int64_t* h_boxes_cls,
                         const double* h_boxes_volume, int32_t n_fg_instances, int32_t instance_classes,
                         int32_t* h_sem_spp, int32_t* h_inst_spp, float* h_prob_spp, float* h_mu_spp,
                         float* h_var_spp) {
  if (!s || !h_boxes_cls || !h_boxes_volume || !h_sem_spp || !h_inst_spp || !h_prob_spp || !h_mu_spp || !h_var_spp)
    return GAPRO_ERR_BAD_ARG;
  if (!s->fits.empty() && (!h_probs_new || !h_labels || !h_mu || !h_var)) return GAPRO_ERR_BAD_ARG;
  const int S = s->n_spps, B = s->n_boxes, W = s->words;
  std::vector<int32_t> inst(S, -100);        // :365
  std::vector<int64_t> determined(S, 0);     // :366
  for (int i = 0; i < S; ++i) {
    h_prob_spp[i] = 0.f;                     // :367
    h_mu_spp[i] = -100.f;                    // :368
    h_var_spp[i] = -100.f;                   // :369
  }
  for (int b = 0; b < B; ++b)
    for (int32_t sp : s->single[b]) {        // :373-377
      inst[sp] = b;
      h_prob_spp[sp] = 1.f;
      determined[sp] = kMaxNum;
    }
  for (int sp = 0; sp < S; ++sp)
    if (s->n_bbs[sp] == 0) {                 // :381-383
      inst[sp] = -1;
      h_prob_spp[sp] = 1.f;
      determined[sp] = kMaxNum;
    }

  int64_t out_off = 0;
  for (const Event& e : s->events) {
    if (e.kind == 0) {                       // :412-414 / :419-421
      for (int32_t sp : e.inter) {
        inst[sp] = e.aux;
        determined[sp] = kMaxNum;
        h_prob_spp[sp] = 1.f;
      }
      continue;
    }
    const int64_t n = (int64_t)e.inter.size();
    for (int64_t j = 0; j < n; ++j) {        // :438-446
      const int32_t sp = e.inter[j];
      const float pn = h_probs_new[out_off + j];
      if (h_prob_spp[sp] < pn) {             // strict, float32
        inst[sp] = h_labels[out_off + j] ? e.b2 : e.b1;
        h_prob_spp[sp] = pn;
        h_mu_spp[sp] = h_mu[out_off + j];
        h_var_spp[sp] = h_var[out_off + j];
        determined[sp] = n;
      }
    }
    out_off += n;
  }

  for (int sp = 0; sp < S; ++sp) {           // :450-463 smallest-volume box, first minimum wins
    if (!(s->n_bbs[sp] > 1 && determined[sp] == 0)) continue;
    double best = std::numeric_limits<double>::infinity();
    int arg = -1;
    for (int w = 0; w < W; ++w) {
      uint64_t bits = s->occ_bits[(size_t)sp * W + w];
      while (bits) {
        const int b = w * 64 + __builtin_ctzll(bits);
        bits &= bits - 1;
        if (h_boxes_volume[b] < best) {
          best = h_boxes_volume[b];
          arg = b;
        }
      }
    }
    if (arg < 0) {  // all volumes +inf/NaN: keep scatter_min's "first" convention
      for (int w = 0; w < W && arg < 0; ++w)
        if (s->occ_bits[(size_t)sp * W + w]) arg = w * 64 + __builtin_ctzll(s->occ_bits[(size_t)sp * W + w]);
    }
    inst[sp] = arg;
    h_prob_spp[sp] = 1.f;
  }

  for (int sp = 0; sp < S; ++sp) {           // :466-476
    int32_t sem = -100, ins = -100;
    if (inst[sp] >= 0) {
      sem = (int32_t)h_boxes_cls[inst[sp]];
      ins = inst[sp];
    } else if (inst[sp] == -1) {
      sem = instance_classes;
    }
    if (ins >= n_fg_instances) ins = -100;   // :474 (after which :475 is a no-op)
    h_sem_spp[sp] = sem;
    h_inst_spp[sp] = ins;
  }
  return GAPRO_OK;
}

}  // extern "C"
