// Native batch feeder of the gen_ps driver: files in -> device memory, device memory -> label files, on threads of its
// own that never touch the Python interpreter.
//
// Replaces the per-scene host half of reference gapro/gen_ps.py:36-132 -- `torch.load` of the scene tuple and the
// superpoint ids (:45-46), the default features from the UN-aligned coordinates (:55), the axis alignment (:58-69),
// getInstanceInfo (:71-77), the upload (:83-87) and, on the way out, the device -> host copies and `torch.save` of the
// 5-tuple (:126-132).  Round 4 ran these steps as Python closures on loader threads (native file decoding inside, the
// orchestration around it under the GIL): beside a running generator they delivered ~250 scenes/s where one MI355X
// takes 370, the first batch left a worker after 2.7 .. 3.2 s of a 5.4 s job, and a worker's pinned staging buffers
// grew scene by scene.  Here the whole chain is one C++ work item per scene:
//
//   load    open the scene / superpoint (/ feature) files (gapro_pth_*), decode the payloads straight into ONE pinned
//           block per scene laid out as the device image  coords f64[N,3] | feats f32[N,D] | spp i64[N] | sem f64[N] |
//           inst f64[N]  (256-byte aligned parts), build the default features, apply the alignment matrix, take the GT
//           instance boxes (gapro_scene_instance_boxes)
//   upload  the caller polls for the scenes that are loaded IN ORDER, hands over a device slab, and gets one
//           asynchronous copy per scene on the feed's copy stream plus an event for the batch
//   export  per scene: five device -> host copies behind the caller's event into a pinned block, gapro_pth_write
//
// Pinned blocks come from a pool (first fit, recycled when a scene has been uploaded / written); the bytes in flight
// are capped, so the readers run ahead of the consumer by a bounded amount.  device < 0 is the host-only mode of
// `gen_ps --dry_run` (pageable blocks, no copies): the same threads, reader and writer without a GPU.
//
// The arithmetic is the reference's, bit for bit: features are float32(xyz | rgb) of the UN-aligned points; the
// alignment is np.dot([x y z 1], A^T) = fma-accumulated in k order (OpenBLAS dgemm's order: tests/test_feeder.py holds
// it to np.dot bitwise); the boxes are getInstanceInfo's min / max per instance id.
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr int64_t kAlign = 256;
inline int64_t up(int64_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

struct Block {
  void* p = nullptr;
  int64_t bytes = 0;
  bool pinned = false;  // page-locked (hipHostRegister) yet?  See pin_block.
};

struct Item {  // one submitted scene
  std::string scene, spp, align, feats;
  gapro_feed_scene rec{};
  std::vector<double> box, cls, vol;
  Block blk;
  int64_t image_bytes = 0;
  int state = 0;  // 0 queued, 1 loading, 2 loaded, 3 handed to an upload
};

struct OutItem {  // one label file to write
  gapro_feed_out o{};
  std::string path;
  hipEvent_t ready = nullptr;
  int64_t seq = 0;  // position in submission order
};

}  // namespace

struct gapro_feed {
  int device = -1;
  int feat_dim_default = 6;
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_work, cv_ready, cv_space, cv_export;
  std::deque<std::unique_ptr<Item>> items;  // in submission order; entries before `base` have been popped
  std::vector<std::unique_ptr<Item>> handed;  // the scenes of the latest upload (their records point into them)
  size_t base = 0;       // index (in submission order) of items.front()
  size_t next_load = 0;  // next scene a loader thread takes
  size_t next_alloc = 0; // next scene allowed to take its staging block (blocks are handed out IN ORDER: scenes leave
                         // in order, so a later scene holding the last room while an earlier one waits would deadlock)
  size_t head = 0;       // next scene the consumer gets
  std::deque<OutItem> exports;
  // label files: exp_done counts finished writes in ANY order (up to 16 writers); exp_prefix is the number of exports
  // that are finished as a CONTIGUOUS prefix of the submission order -- the only count a caller may release device
  // memory by (item k + 1 finishing says nothing about item k still inside its device -> host copies)
  int64_t exp_submitted = 0, exp_done = 0, exp_failed = 0, exp_prefix = 0;
  std::deque<char> exp_flags;  // finished flags of exports exp_prefix, exp_prefix + 1, ...
  int exp_active = 0;          // writers between taking an export and finishing it
  int budget_waiters = 0;      // loaders whose turn it is and whom only the byte budget holds back
  bool failed = false;         // an upload failed half way: the feed's order is gone, every later call reports it
  // where the loaders' time goes (gapro_feed_stats; GAPRO_DRIVER_TIMES=1 prints it)
  std::atomic<int64_t> st_pin_us{0}, st_pin_n{0}, st_pin_bytes{0}, st_load_us{0}, st_load_n{0}, st_wait_us{0}, st_write_us{0}, st_write_n{0};
  std::chrono::steady_clock::time_point t_create = std::chrono::steady_clock::now();
  std::atomic<int64_t> st_first_loaded_us{-1};
  std::vector<std::string> exp_errors;
  bool stop = false, closed = false;
  // pinned block pool
  std::vector<Block> free_blocks;
  int64_t inflight_bytes = 0, budget_bytes = 0, pool_bytes = 0;
  // uploads
  hipStream_t copy_stream = nullptr;
  struct Batch {
    int64_t id;
    hipEvent_t ev;
    std::vector<Block> blocks;
    int64_t bytes;
  };
  std::deque<Batch> batches;
  int64_t next_batch_id = 1;
  std::string last_error;
};

namespace {

bool use_gpu(const gapro_feed* f) { return f->device >= 0; }

// turn < 0: a label file (any time); otherwise the scene's index in submission order: its turn comes when every earlier
// scene has taken (or given up) its block
void reap_batches(gapro_feed* f, bool wait_all);

void release_block(const Block& b) {
  if (!b.p) return;
  if (b.pinned) (void)hipHostUnregister(b.p);
  free(b.p);
}

Block take_block(gapro_feed* f, int64_t need, std::unique_lock<std::mutex>& lk, long long turn = -1) {
  // wait for room in the in-flight budget (one scene is always admitted), then first fit from the pool.  The pinned
  // blocks of uploaded batches come back through reap_batches, which round 5 ran on the consumer's thread only: a
  // consumer that was itself waiting (gapro_feed_poll for a batch larger than the budget, the final
  // gapro_feed_export_wait) never reaped, and both sides waited for ever.  So a waiter reaps by itself (a
  // timed wait: completed copies raise no condition variable), and a loader that only the budget holds back says so
  // to the poller, which then hands out what is loaded instead of waiting for a count that cannot be reached.
  bool counted = false;
  for (;;) {
    if (f->stop) break;
    if (use_gpu(f)) reap_batches(f, false);
    const bool my_turn = turn < 0 || (size_t)turn == f->next_alloc;
    // a label file (turn < 0) is always admitted: at most one per thread is in flight, its block comes back by itself,
    // and a writer waiting for room that only loaded-but-untaken scenes hold -- while the poller waits for the
    // writers -- would be the same deadlock from the other side
    const bool room = turn < 0 || f->inflight_bytes == 0 || f->inflight_bytes + need <= f->budget_bytes;
    if (my_turn && room) break;
    if (turn >= 0 && my_turn && !counted) {
      counted = true;
      ++f->budget_waiters;
      f->cv_ready.notify_all();
    }
    f->cv_space.wait_for(lk, std::chrono::milliseconds(2));
  }
  if (counted) --f->budget_waiters;
  Block b;
  if (f->stop) return b;
  f->inflight_bytes += need;
  if (turn >= 0) {
    ++f->next_alloc;
    f->cv_space.notify_all();
  }
  size_t best = f->free_blocks.size();
  for (size_t i = 0; i < f->free_blocks.size(); ++i)
    if (f->free_blocks[i].bytes >= need && (best == f->free_blocks.size() || f->free_blocks[i].bytes < f->free_blocks[best].bytes))
      best = i;
  if (best < f->free_blocks.size()) {
    b = f->free_blocks[best];
    f->free_blocks.erase(f->free_blocks.begin() + best);
    return b;
  }
  // a pool that has outgrown its budget gives its smallest blocks back first
  while (!f->free_blocks.empty() && f->pool_bytes + need > 2 * f->budget_bytes) {
    size_t sm = 0;
    for (size_t i = 1; i < f->free_blocks.size(); ++i)
      if (f->free_blocks[i].bytes < f->free_blocks[sm].bytes) sm = i;
    Block d = f->free_blocks[sm];
    f->free_blocks.erase(f->free_blocks.begin() + sm);
    f->pool_bytes -= d.bytes;
    lk.unlock();
    release_block(d);
    lk.lock();
  }
  const int64_t sz = (std::max<int64_t>(need, 4 << 20) + (2 << 20) - 1) / (2 << 20) * (2 << 20);
  f->pool_bytes += sz;
  lk.unlock();  // the allocation itself (page pinning) runs outside the lock, in parallel on the loader threads
  // Plain memory now, page-locked later by whoever has filled it (pin_block).  Round 5 took staging blocks from
  // hipHostMalloc, which allocates, clears and pins at 6 .. 7 GB/s whatever the block size or the number of calling
  // threads (tools/probes/pin_rate.hip): the loaders of a fresh worker spent 58 % of their first second inside it -- 14 s
  // of thread time for 5.6 GB -- and that, not the disk, set the pace at which the first batches arrived.  Registering
  // memory that is already touched runs at ~100 GB/s, and the touching is the decoder's own writes, on 15 threads at once.
  void* p = nullptr;
  if (posix_memalign(&p, 2 << 20, (size_t)sz) != 0) p = nullptr;
  else (void)madvise(p, (size_t)sz, MADV_HUGEPAGE);
  f->st_pin_n += 1;
  f->st_pin_bytes += sz;
  lk.lock();
  if (!p) {
    f->pool_bytes -= sz;
    f->inflight_bytes -= need;
    return b;
  }
  b.p = p;
  b.bytes = sz;
  return b;
}

// page-lock a block once its pages exist (the caller has just filled it; what is left untouched -- the rounding tail, a
// fresh label block -- is faulted in by the registration itself)
bool pin_block(gapro_feed* f, Block& b) {
  if (!use_gpu(f) || b.pinned || !b.p) return true;
  const auto tp0 = std::chrono::steady_clock::now();
  const bool ok = hipHostRegister(b.p, (size_t)b.bytes, hipHostRegisterDefault) == hipSuccess;
  if (!ok) (void)hipGetLastError();  // the copies still work from pageable memory, only slower
  b.pinned = ok;
  f->st_pin_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tp0).count();
  return ok;
}

void give_block(gapro_feed* f, Block b, int64_t need) {  // lock held
  if (b.p) f->free_blocks.push_back(b);
  f->inflight_bytes -= need;
  f->cv_space.notify_all();
}

// 'axisAlignment = a00 ... a33' (gen_ps.py:58-64): the sixteen numbers behind the '=' of the first line naming it
bool read_alignment(const std::string& path, double A[16]) {
  FILE* fh = fopen(path.c_str(), "r");
  if (!fh) return false;
  char line[4096];
  bool ok = false;
  while (fgets(line, sizeof(line), fh)) {
    if (!strstr(line, "axisAlignment")) continue;
    const char* p = strchr(line, '=');
    if (!p) break;
    ++p;
    int k = 0;
    for (; k < 16; ++k) {
      char* e = nullptr;
      A[k] = strtod(p, &e);
      if (e == p) break;
      p = e;
    }
    ok = k == 16;
    break;
  }
  fclose(fh);
  return ok;
}

// read array `idx` of a file as float64 into dst (the file holds float64 or float32)
int read_f64(gapro_pth_file* f, int idx, int64_t n_elems, double* dst, std::vector<char>& tmp) {
  gapro_pth_array a;
  if (gapro_pth_info(f, idx, &a) != GAPRO_OK) return GAPRO_ERR_IO;
  if (a.kind != 'f' || a.nbytes != n_elems * a.itemsize) return GAPRO_ERR_UNSUPPORTED;
  if (a.itemsize == 8) return gapro_pth_read(f, idx, dst, a.nbytes);
  if (a.itemsize != 4) return GAPRO_ERR_UNSUPPORTED;
  tmp.resize((size_t)a.nbytes);
  const int rc = gapro_pth_read(f, idx, tmp.data(), a.nbytes);
  if (rc != GAPRO_OK) return rc;
  const float* s = (const float*)tmp.data();
  for (int64_t i = 0; i < n_elems; ++i) dst[i] = (double)s[i];
  return GAPRO_OK;
}

// The host half of a scene (gen_ps.py:37-77) into a block of the pool.
void load_scene(gapro_feed* f, size_t index, Item& it, std::vector<char>& tmp, std::vector<double>& rgb) {
  gapro_pth_file* fs = nullptr;
  gapro_pth_file* fp = nullptr;
  gapro_pth_file* ff = nullptr;
  bool had_turn = false;
  auto done = [&](int code) {
    if (fs) gapro_pth_close(fs);
    if (fp) gapro_pth_close(fp);
    if (ff) gapro_pth_close(ff);
    it.rec.status = code;
    if (!had_turn) {  // a scene that fails before it needs memory still passes its turn on
      std::unique_lock<std::mutex> lk(f->mu);
      f->cv_space.wait(lk, [&] { return f->stop || index == f->next_alloc; });
      if (!f->stop) ++f->next_alloc;
      f->cv_space.notify_all();
    }
  };
  int rc = gapro_pth_open(it.scene.c_str(), &fs);
  if (rc != GAPRO_OK) return done(rc);
  gapro_pth_array ax;
  if (gapro_pth_count(fs) != 4 || gapro_pth_info(fs, 0, &ax) != GAPRO_OK || ax.ndim != 2 || ax.shape[1] != 3 ||
      ax.kind != 'f')
    return done(GAPRO_ERR_UNSUPPORTED);  // not the (xyz, rgb, sem, inst) tuple of prepare_data_inst.py:104
  const int64_t N = ax.shape[0];
  rc = gapro_pth_open(it.spp.c_str(), &fp);
  if (rc != GAPRO_OK) return done(rc);
  gapro_pth_array as;
  if (gapro_pth_count(fp) != 1 || gapro_pth_info(fp, 0, &as) != GAPRO_OK || as.kind != 'i' ||
      as.nbytes != N * as.itemsize || (as.itemsize != 8 && as.itemsize != 4))
    return done(GAPRO_ERR_UNSUPPORTED);
  int64_t D = f->feat_dim_default;
  gapro_pth_array af;
  if (!it.feats.empty()) {  // --use_deepfeat: f32[N,D] written by isbnet.py:512-515
    rc = gapro_pth_open(it.feats.c_str(), &ff);
    if (rc != GAPRO_OK) return done(rc);
    if (gapro_pth_count(ff) != 1 || gapro_pth_info(ff, 0, &af) != GAPRO_OK || af.kind != 'f' || af.ndim != 2 ||
        af.shape[0] != N || (af.itemsize != 4 && af.itemsize != 8))
      return done(GAPRO_ERR_UNSUPPORTED);
    D = af.shape[1];
  }
  double A[16];
  if (!read_alignment(it.align, A)) return done(GAPRO_ERR_IO);
  // device image of the scene
  gapro_feed_scene& r = it.rec;
  r.n_points = (int32_t)N;
  r.feat_dim = (int32_t)D;
  r.off_coords = 0;
  r.off_feats = up(24 * N);
  r.off_spp = r.off_feats + up(4 * D * N);
  r.off_sem = r.off_spp + up(8 * N);
  r.off_inst = r.off_sem + up(8 * N);
  it.image_bytes = r.off_inst + up(8 * N);
  {
    std::unique_lock<std::mutex> lk(f->mu);
    it.blk = take_block(f, it.image_bytes, lk, (long long)index);
    had_turn = !f->stop;
  }
  if (!it.blk.p) return done(f->stop ? GAPRO_ERR_BAD_ARG : GAPRO_ERR_OOM);
  char* base = (char*)it.blk.p;
  double* xyz = (double*)(base + r.off_coords);
  float* feats = (float*)(base + r.off_feats);
  int64_t* spp = (int64_t*)(base + r.off_spp);
  double* sem = (double*)(base + r.off_sem);
  double* inst = (double*)(base + r.off_inst);
  if ((rc = read_f64(fs, 0, 3 * N, xyz, tmp)) != GAPRO_OK) return done(rc);
  if ((rc = read_f64(fs, 2, N, sem, tmp)) != GAPRO_OK) return done(rc);
  if ((rc = read_f64(fs, 3, N, inst, tmp)) != GAPRO_OK) return done(rc);
  if (as.itemsize == 8) {
    if ((rc = gapro_pth_read(fp, 0, spp, as.nbytes)) != GAPRO_OK) return done(rc);
  } else {
    tmp.resize((size_t)as.nbytes);
    if ((rc = gapro_pth_read(fp, 0, tmp.data(), as.nbytes)) != GAPRO_OK) return done(rc);
    const int32_t* s = (const int32_t*)tmp.data();
    for (int64_t i = 0; i < N; ++i) spp[i] = s[i];
  }
  if (ff) {
    if (af.itemsize == 4) {
      if ((rc = gapro_pth_read(ff, 0, feats, af.nbytes)) != GAPRO_OK) return done(rc);
    } else {
      tmp.resize((size_t)af.nbytes);
      if ((rc = gapro_pth_read(ff, 0, tmp.data(), af.nbytes)) != GAPRO_OK) return done(rc);
      const double* s = (const double*)tmp.data();
      for (int64_t i = 0; i < N * D; ++i) feats[i] = (float)s[i];
    }
  } else {
    // np.concatenate([xyz, rgb], -1) of the UN-aligned coordinates, as float32 (:55, :84)
    rgb.resize((size_t)(3 * N));
    if ((rc = read_f64(fs, 1, 3 * N, rgb.data(), tmp)) != GAPRO_OK) return done(rc);
    gapro_scene_default_feats(xyz, rgb.data(), N, feats);
  }
  // pts = [x y z 1];  np.dot(pts, A^T)[:, :3]  (:65-69): row j of A against the point, accumulated in k order
  for (int64_t i = 0; i < N; ++i) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    for (int j = 0; j < 3; ++j) {
      double acc = x * A[4 * j];
      acc = std::fma(y, A[4 * j + 1], acc);
      acc = std::fma(z, A[4 * j + 2], acc);
      acc = std::fma(1.0, A[4 * j + 3], acc);
      xyz[3 * i + j] = acc;
    }
  }
  // getInstanceInfo on the aligned points (:71-77)
  int32_t nb = 0, inum = 0;
  int cap = 256;
  for (;;) {
    it.box.resize(6 * (size_t)cap);
    it.cls.resize((size_t)cap);
    it.vol.resize((size_t)cap);
    rc = gapro_scene_instance_boxes(xyz, inst, sem, N, 1, cap, it.box.data(), it.cls.data(), it.vol.data(), &nb, &inum);
    if (rc == GAPRO_OK) break;
    if (inum > cap) {
      cap = inum;
      continue;
    }
    return done(rc);
  }
  r.n_instances = nb;
  r.inst_box = it.box.data();
  r.inst_cls = it.cls.data();
  r.inst_vol = it.vol.data();
  (void)pin_block(f, it.blk);  // (a recycled block is pinned already; a fresh one: every page of the image is touched)
  done(GAPRO_OK);
}

void write_labels(gapro_feed* f, OutItem& w, hipStream_t st, std::string* err) {
  const gapro_feed_out& o = w.o;
  const int64_t n = o.n_points, s = o.n_mu;
  const int64_t off1 = up(4 * n), off2 = 2 * off1, off3 = 3 * off1, off4 = off3 + up(4 * s);
  const int64_t need = off4 + up(4 * s);
  Block b;
  {
    std::unique_lock<std::mutex> lk(f->mu);
    b = take_block(f, need, lk);
  }
  if (!b.p) {
    *err = w.path + ": no staging memory";
    return;
  }
  char* h = (char*)b.p;
  bool ok = true;
  (void)pin_block(f, b);
  if (use_gpu(f)) {
    if (w.ready) ok = hipStreamWaitEvent(st, w.ready, 0) == hipSuccess;
    ok = ok && hipMemcpyAsync(h, o.d_sem, 4 * n, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(h + off1, o.d_inst, 4 * n, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(h + off2, o.d_prob, 4 * n, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(h + off3, o.d_mu, 4 * s, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(h + off4, o.d_var, 4 * s, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess;
  } else {  // host-only mode: the "device" pointers are host memory
    memcpy(h, o.d_sem, 4 * n);
    memcpy(h + off1, o.d_inst, 4 * n);
    memcpy(h + off2, o.d_prob, 4 * n);
    memcpy(h + off3, o.d_mu, 4 * s);
    memcpy(h + off4, o.d_var, 4 * s);
  }
  if (!ok) {
    *err = w.path + ": device -> host copy failed";
  } else {
    gapro_pth_array d[5];
    memset(d, 0, sizeof(d));
    const int64_t len[5] = {n, n, n, s, s};
    const void* data[5] = {h, h + off1, h + off2, h + off3, h + off4};
    for (int k = 0; k < 5; ++k) {
      d[k].kind = k < 2 ? 'i' : 'f';
      d[k].itemsize = 4;
      d[k].ndim = 1;
      d[k].shape[0] = len[k];
      d[k].shape[1] = d[k].shape[2] = d[k].shape[3] = 1;
      d[k].nbytes = 4 * len[k];
    }
    if (gapro_pth_write(w.path.c_str(), 5, d, data, 1) != GAPRO_OK) *err = w.path + ": " + gapro_pth_last_error();
  }
  std::lock_guard<std::mutex> lk(f->mu);
  give_block(f, b, need);
}

void worker(gapro_feed* f) {
  hipStream_t st = nullptr;
  if (use_gpu(f)) {
    (void)hipSetDevice(f->device);
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  }
  std::vector<char> tmp;
  std::vector<double> rgb;
  std::unique_lock<std::mutex> lk(f->mu);
  for (;;) {
    // Label files first: they free device memory and are what the run is measured by -- but only files whose arrays
    // are READY.  Round 5 took an export as soon as it was queued and then sat in hipStreamSynchronize behind the
    // caller's event, i.e. behind the broadcast kernels of a batch that wait for the running fit launch to drain: with a
    // batch of 256 files queued, all 15 threads parked there for 0.3 .. 0.6 s per batch (16 ms per file of thread time for
    // 1.4 ms of work: gapro_feed_stats) and nothing was READ meanwhile -- which is why the reads ran at 250 scenes/s beside
    // a running generator and at 850 without one.  Now a thread looks at the front export's event and, while it has not
    // completed, loads scenes instead (or naps 1 ms when there is nothing to load).
    bool export_ready = false;
    for (;;) {
      if (f->stop) break;
      export_ready = false;
      if (!f->exports.empty()) {
        const hipEvent_t ev = f->exports.front().ready;
        export_ready = !use_gpu(f) || !ev || hipEventQuery(ev) != hipErrorNotReady;
      }
      const bool can_load = f->next_load < f->base + f->items.size();
      if (export_ready || can_load) break;
      if (!f->exports.empty()) f->cv_work.wait_for(lk, std::chrono::milliseconds(1));  // an event completes silently
      else f->cv_work.wait(lk);
    }
    if (f->stop) break;
    if (export_ready) {
      OutItem w = std::move(f->exports.front());
      f->exports.pop_front();
      ++f->exp_active;
      lk.unlock();
      std::string err;
      const auto tw0 = std::chrono::steady_clock::now();
      write_labels(f, w, st, &err);
      f->st_write_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tw0).count();
      f->st_write_n += 1;
      lk.lock();
      --f->exp_active;
      ++f->exp_done;
      f->exp_flags[(size_t)(w.seq - f->exp_prefix)] = 1;
      while (!f->exp_flags.empty() && f->exp_flags.front()) {
        f->exp_flags.pop_front();
        ++f->exp_prefix;
      }
      if (!err.empty()) {
        ++f->exp_failed;
        f->exp_errors.push_back(err);
      }
      f->cv_export.notify_all();
      f->cv_ready.notify_all();
      continue;
    }
    const size_t i = f->next_load++;
    Item& it = *f->items[i - f->base];
    it.state = 1;
    lk.unlock();
    const auto tl0 = std::chrono::steady_clock::now();
    load_scene(f, i, it, tmp, rgb);
    const auto tl1 = std::chrono::steady_clock::now();
    f->st_load_us += std::chrono::duration_cast<std::chrono::microseconds>(tl1 - tl0).count();
    f->st_load_n += 1;
    if (i == 0) f->st_first_loaded_us = std::chrono::duration_cast<std::chrono::microseconds>(tl1 - f->t_create).count();
    lk.lock();
    if (it.rec.status != GAPRO_OK && it.blk.p) {  // a failed scene holds no staging memory
      give_block(f, it.blk, it.image_bytes);
      it.blk = Block();
    }
    it.state = 2;
    f->cv_ready.notify_all();
  }
  lk.unlock();
  if (st) (void)hipStreamDestroy(st);
}

// batches whose copies have completed give their pinned blocks back (lock held)
void reap_batches(gapro_feed* f, bool wait_all) {
  while (!f->batches.empty()) {
    gapro_feed::Batch& b = f->batches.front();
    if (b.ev) {
      const hipError_t e = wait_all ? hipEventSynchronize(b.ev) : hipEventQuery(b.ev);
      if (e == hipErrorNotReady) break;
      (void)hipEventDestroy(b.ev);
    }
    for (Block& blk : b.blocks)
      if (blk.p) f->free_blocks.push_back(blk);
    f->inflight_bytes -= b.bytes;
    f->batches.pop_front();
    f->cv_space.notify_all();
  }
}

}  // namespace

extern "C" {

int gapro_feed_create(int32_t device, int32_t n_threads, int64_t budget_bytes, int32_t default_feat_dim,
                      gapro_feed** out) {
  if (!out || n_threads <= 0 || n_threads > 256 || budget_bytes <= 0 || default_feat_dim <= 0) return GAPRO_ERR_BAD_ARG;
  *out = nullptr;
  gapro_feed* f = new (std::nothrow) gapro_feed();
  if (!f) return GAPRO_ERR_OOM;
  f->device = device;
  f->budget_bytes = budget_bytes;
  f->feat_dim_default = default_feat_dim;
  if (use_gpu(f)) {
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking) != hipSuccess) {
      delete f;
      return GAPRO_ERR_HIP;
    }
  }
  for (int i = 0; i < n_threads; ++i) f->threads.emplace_back(worker, f);
  *out = f;
  return GAPRO_OK;
}

static void feed_stop_threads(gapro_feed* f) {
  {
    std::lock_guard<std::mutex> lk(f->mu);
    f->stop = true;
  }
  f->cv_work.notify_all();
  f->cv_space.notify_all();
  f->cv_ready.notify_all();
  f->cv_export.notify_all();
  for (std::thread& t : f->threads) t.join();
  f->threads.clear();
}

void gapro_feed_detach(gapro_feed* f) {
  if (f) feed_stop_threads(f);  // the staging memory (and this object) stay until the process ends
}

void gapro_feed_destroy(gapro_feed* f) {
  if (!f) return;
  feed_stop_threads(f);
  {
    std::lock_guard<std::mutex> lk(f->mu);
    if (use_gpu(f)) reap_batches(f, true);
    for (auto& it : f->items)
      if (it->blk.p) f->free_blocks.push_back(it->blk);
    for (Block& b : f->free_blocks) release_block(b);
  }
  if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
  delete f;
}

const char* gapro_feed_last_error(const gapro_feed* f) { return f ? f->last_error.c_str() : "null feed"; }

int gapro_feed_submit(gapro_feed* f, int32_t n, const char* const* scene_paths, const char* const* spp_paths,
                      const char* const* align_paths, const char* const* feat_paths) {
  if (!f || n < 0 || (n > 0 && (!scene_paths || !spp_paths || !align_paths))) return GAPRO_ERR_BAD_ARG;
  {
    std::lock_guard<std::mutex> lk(f->mu);
    if (f->closed) return GAPRO_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i) {
      std::unique_ptr<Item> it(new Item());
      it->scene = scene_paths[i];
      it->spp = spp_paths[i];
      it->align = align_paths[i];
      if (feat_paths && feat_paths[i]) it->feats = feat_paths[i];
      f->items.push_back(std::move(it));
    }
  }
  f->cv_work.notify_all();
  return GAPRO_OK;
}

int gapro_feed_close(gapro_feed* f) {
  if (!f) return GAPRO_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(f->mu);
  f->closed = true;
  f->cv_ready.notify_all();
  return GAPRO_OK;
}

int gapro_feed_poll(gapro_feed* f, int32_t min_ready, int32_t max_scenes, int32_t timeout_ms, int32_t* n_ready,
                    int64_t* slab_bytes) {
  if (!f || !n_ready || !slab_bytes || max_scenes <= 0) return GAPRO_ERR_BAD_ARG;
  std::unique_lock<std::mutex> lk(f->mu);
  if (f->failed) return GAPRO_ERR_HIP;
  auto count = [&](int64_t* bytes) {
    int n = 0;
    int64_t b = 0;
    for (size_t i = f->head; i < f->base + f->items.size() && n < max_scenes; ++i) {
      const Item& it = *f->items[i - f->base];
      if (it.state != 2) break;
      b += it.rec.status == GAPRO_OK ? it.image_bytes : 0;
      ++n;
    }
    if (bytes) *bytes = b;
    return n;
  };
  const size_t want_cap = (size_t)std::min<int64_t>(max_scenes, std::max(min_ready, 1));
  auto enough = [&] {
    if (f->stop) return true;
    const size_t submitted = f->base + f->items.size() - f->head;  // scenes not yet handed out
    const size_t want = f->closed ? std::min(want_cap, submitted) : want_cap;
    const size_t have = (size_t)count(nullptr);
    if (have >= want || (f->closed && submitted == 0)) return true;
    // The byte budget is smaller than what `want` scenes need (deep features, --batch_scenes 512, S3DIS
    // rooms): the loader whose turn it is waits for room that only THIS caller can make, by taking what is loaded.
    // Nothing else will return memory: no upload is in flight (its blocks come back when its copy is done) and no label
    // file is being written (its block comes back when it is on disk).  Label files that are QUEUED do not count: with
    // every thread parked as a loader behind the budget nobody is left to write them, and waiting for them here was a
    // second, rarer form of the same deadlock (one run in six of the budget test under CPU load).
    return have >= 1 && f->budget_waiters > 0 && f->batches.empty() && f->exp_active == 0;
  };
  // (a timed wait: uploads whose copies complete raise no condition variable, and their blocks are what the loaders wait for)
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
  for (;;) {
    if (use_gpu(f)) reap_batches(f, false);
    if (enough()) break;
    auto slice = std::chrono::milliseconds(f->batches.empty() ? 50 : 2);
    if (timeout_ms >= 0) {
      const auto now = std::chrono::steady_clock::now();
      if (now >= t_end) break;
      slice = std::min(slice, std::chrono::duration_cast<std::chrono::milliseconds>(t_end - now) + std::chrono::milliseconds(1));
    }
    f->cv_ready.wait_for(lk, slice);
  }
  *n_ready = count(slab_bytes);
  return GAPRO_OK;
}

int gapro_feed_upload(gapro_feed* f, int32_t n, void* d_slab, int64_t slab_bytes, void* slab_stream, gapro_feed_scene* out,
                      int64_t* batch_id) {
  if (!f || n <= 0 || !out || !batch_id || (use_gpu(f) && (!d_slab || slab_bytes <= 0))) return GAPRO_ERR_BAD_ARG;
  std::unique_lock<std::mutex> lk(f->mu);
  if (f->failed) {
    f->last_error = "gapro_feed_upload: an earlier upload failed half way; the feed cannot continue";
    return GAPRO_ERR_HIP;
  }
  // everything is checked BEFORE any state changes (an error return from the middle of the loop left
  // scenes marked as handed out, their pinned blocks dropped and the caller's name list out of step)
  if (f->head + n > f->base + f->items.size()) return GAPRO_ERR_BAD_ARG;
  int64_t total = 0;
  for (int k = 0; k < n; ++k) {
    const Item& it = *f->items[f->head + k - f->base];
    if (it.state != 2) return GAPRO_ERR_BAD_ARG;
    if (it.rec.status == GAPRO_OK) total += it.image_bytes;
  }
  if (use_gpu(f) && total > slab_bytes) {
    f->last_error = "gapro_feed_upload: the device slab is smaller than the size gapro_feed_poll reported";
    return GAPRO_ERR_BAD_ARG;
  }
  gapro_feed::Batch b;
  b.id = f->next_batch_id++;
  b.ev = nullptr;
  b.bytes = 0;
  bool hip_ok = true;
  if (use_gpu(f) && slab_stream != f->copy_stream) {
    // The slab may come from a caching allocator (torch's): a block whose previous owner was freed on the host while
    // its kernels are still queued on the allocating stream.  The copies below run on the feed's own stream, which
    // nothing orders behind that work, so they wait here for everything queued on the caller's stream so far (ADVICE
    // r05).  slab_stream = NULL is the legacy default stream.
    hipEvent_t e = nullptr;
    hip_ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess &&
             hipEventRecord(e, (hipStream_t)slab_stream) == hipSuccess &&
             hipStreamWaitEvent(f->copy_stream, e, 0) == hipSuccess;
    if (e) (void)hipEventDestroy(e);  // (destroying a recorded event is deferred by the runtime until it completes)
  }
  int64_t off = 0;
  for (int k = 0; k < n; ++k) {
    Item& it = *f->items[f->head + k - f->base];
    out[k] = it.rec;
    it.state = 3;
    if (it.rec.status != GAPRO_OK) continue;
    if (use_gpu(f)) {
      hip_ok = hip_ok && hipMemcpyAsync((char*)d_slab + off, it.blk.p, (size_t)it.image_bytes, hipMemcpyHostToDevice,
                                        f->copy_stream) == hipSuccess;
      out[k].off_coords += off;
      out[k].off_feats += off;
      out[k].off_spp += off;
      out[k].off_sem += off;
      out[k].off_inst += off;
      off += it.image_bytes;
    } else {  // host-only mode: the offsets are relative to the scene's own block, whose address travels in host_image
      out[k].host_image = it.blk.p;
    }
    b.blocks.push_back(it.blk);
    b.bytes += it.image_bytes;
    it.blk = Block();
  }
  f->head += n;
  // scenes that have been handed out leave the list; the box arrays their records point to live until the NEXT upload
  f->handed.clear();
  while (f->base < f->head) {
    f->handed.push_back(std::move(f->items.front()));
    f->items.pop_front();
    ++f->base;
  }
  if (use_gpu(f)) {
    hip_ok = hip_ok && hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) == hipSuccess &&
             hipEventRecord(b.ev, f->copy_stream) == hipSuccess;
    if (!hip_ok) {
      // the copies that were queued may still run: wait for them, then the blocks go back through the usual path
      (void)hipStreamSynchronize(f->copy_stream);
      if (b.ev) (void)hipEventDestroy(b.ev);
      b.ev = nullptr;
    }
  }
  *batch_id = b.id;
  f->batches.push_back(std::move(b));  // (host-only mode: the blocks stay with the batch until gapro_feed_release_batch)
  if (!hip_ok) {
    f->failed = true;
    reap_batches(f, false);
    f->last_error = "gapro_feed_upload: a HIP call failed (event or host -> device copy); the feed stops here";
    return GAPRO_ERR_HIP;
  }
  return GAPRO_OK;
}

int gapro_feed_batch_wait(gapro_feed* f, int64_t batch_id, void* stream) {
  if (!f) return GAPRO_ERR_BAD_ARG;
  if (!use_gpu(f)) return GAPRO_OK;
  std::lock_guard<std::mutex> lk(f->mu);
  for (gapro_feed::Batch& b : f->batches)
    if (b.id == batch_id) {
      if (hipStreamWaitEvent((hipStream_t)stream, b.ev, 0) != hipSuccess) return GAPRO_ERR_HIP;
      return GAPRO_OK;
    }
  return GAPRO_OK;  // already complete and reaped
}

int gapro_feed_release_batch(gapro_feed* f, int64_t batch_id) {
  if (!f) return GAPRO_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(f->mu);
  if (use_gpu(f)) {
    reap_batches(f, false);
    return GAPRO_OK;
  }
  for (size_t i = 0; i < f->batches.size(); ++i)
    if (f->batches[i].id == batch_id) {
      for (Block& blk : f->batches[i].blocks)
        if (blk.p) f->free_blocks.push_back(blk);
      f->inflight_bytes -= f->batches[i].bytes;
      f->batches.erase(f->batches.begin() + i);
      f->cv_space.notify_all();
      break;
    }
  return GAPRO_OK;
}

int gapro_feed_export(gapro_feed* f, int32_t n, const gapro_feed_out* items, void* ready_event) {
  if (!f || n < 0 || (n > 0 && !items)) return GAPRO_ERR_BAD_ARG;
  {
    std::lock_guard<std::mutex> lk(f->mu);
    for (int i = 0; i < n; ++i) {
      if (!items[i].path || items[i].n_points < 0 || items[i].n_mu < 0) return GAPRO_ERR_BAD_ARG;
      OutItem w;
      w.o = items[i];
      w.path = items[i].path;
      w.o.path = nullptr;
      w.ready = (hipEvent_t)ready_event;
      w.seq = f->exp_submitted;
      f->exports.push_back(std::move(w));
      f->exp_flags.push_back(0);
      ++f->exp_submitted;
    }
  }
  f->cv_work.notify_all();
  return GAPRO_OK;
}

int gapro_feed_export_wait(gapro_feed* f, int64_t until_done, int32_t timeout_ms, int64_t* n_done, int64_t* n_failed) {
  if (!f) return GAPRO_ERR_BAD_ARG;
  std::unique_lock<std::mutex> lk(f->mu);
  const int64_t target = until_done < 0 ? f->exp_submitted : std::min(until_done, f->exp_submitted);
  // n_done = exports finished as a contiguous prefix of the submission order (see exp_prefix)
  auto ok = [&] { return f->stop || f->exp_prefix >= target; };
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
  for (;;) {
    // the writers wait for staging blocks that the last uploads still hold: nobody else is left to reap them
    if (use_gpu(f)) reap_batches(f, false);
    if (ok()) break;
    auto slice = std::chrono::milliseconds(f->batches.empty() ? 50 : 2);
    if (timeout_ms >= 0) {
      const auto now = std::chrono::steady_clock::now();
      if (now >= t_end) break;
      slice = std::min(slice, std::chrono::duration_cast<std::chrono::milliseconds>(t_end - now) + std::chrono::milliseconds(1));
    }
    f->cv_export.wait_for(lk, slice);
  }
  if (n_done) *n_done = f->exp_prefix;
  if (n_failed) *n_failed = f->exp_failed;
  return GAPRO_OK;
}

int gapro_feed_stats(gapro_feed* f, double* out8) {
  if (!f || !out8) return GAPRO_ERR_BAD_ARG;
  out8[0] = 1e-6 * (double)f->st_pin_us;      // seconds inside hipHostRegister, summed over threads
  out8[1] = (double)f->st_pin_n;              // blocks allocated
  out8[2] = (double)f->st_pin_bytes;          // bytes allocated
  out8[3] = 1e-6 * (double)f->st_load_us;     // seconds inside load_scene (incl. the waits for a block), summed
  out8[4] = (double)f->st_load_n;
  out8[5] = 1e-6 * (double)f->st_write_us;    // seconds inside write_labels, summed
  out8[6] = (double)f->st_write_n;
  out8[7] = 1e-6 * (double)f->st_first_loaded_us;  // first scene loaded, seconds after gapro_feed_create (< 0: none yet)
  return GAPRO_OK;
}

int gapro_feed_export_error(gapro_feed* f, int32_t index, char* buf, int32_t cap) {
  if (!f || !buf || cap <= 0) return GAPRO_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(f->mu);
  if (index < 0 || (size_t)index >= f->exp_errors.size()) return GAPRO_ERR_BAD_ARG;
  snprintf(buf, (size_t)cap, "%s", f->exp_errors[index].c_str());
  return GAPRO_OK;
}

}  // extern "C"
