#!/usr/bin/env python3
"""What the host of this box gives a process: CPU quota (cgroup), and how pure compute / memory copies scale with
threads -- the yardstick for tools/host_ceiling.py (is a flat scenes/s curve the driver's fault or the box's?)."""
import ctypes, os, subprocess, sys, time, threading
import numpy as np

def rd(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return "n/a (%s)" % e.__class__.__name__

print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/memory.max", "/proc/loadavg"):
    print(p, "=", rd(p))
print(subprocess.run("df -h /tmp /dev/shm . | cat; mount | grep -E ' /tmp | / ' | head -3", shell=True, capture_output=True, text=True).stdout)

def spin(n_threads, secs=1.0):
    """numpy releases the GIL in large ufuncs: threads of sqrt over a private 4 MB array (L2-resident compute)"""
    done = [0] * n_threads
    stop = time.time() + secs
    def w(i):
        a = np.random.rand(1 << 19)
        while time.time() < stop:
            np.sqrt(a, out=a); np.add(a, 1.0, out=a)
            done[i] += 1
    ts = [threading.Thread(target=w, args=(i,)) for i in range(n_threads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    return sum(done) / secs

def copy(n_threads, secs=1.0):
    done = [0] * n_threads
    stop = time.time() + secs
    def w(i):
        a = np.random.rand(1 << 22); b = np.empty_like(a)   # 32 MB each: memory-bound
        while time.time() < stop:
            np.copyto(b, a)
            done[i] += 1
    ts = [threading.Thread(target=w, args=(i,)) for i in range(n_threads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    return sum(done) * 64e6 / secs / 1e9

base = None
for n in (1, 2, 4, 8, 16, 32, 64, 128):
    r = spin(n)
    base = base or r
    print("compute threads %3d: %8.0f iters/s  = %5.1f x one thread   copy: %6.1f GB/s" % (n, r, r / base, copy(n)), flush=True)
