// How fast does this runtime pin host memory?  (round 6: the feeder's loaders spend 58 % of their time inside
// hipHostMalloc during a worker's first second -- tools: gapro_feed_stats.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pin_rate tools/probes/pin_rate.hip -lpthread && /tmp/pin_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipSetDevice(0);
  void* d = nullptr;
  hipMalloc(&d, 1ull << 30);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  // 1. one thread, block sizes
  for (size_t mb : {14, 64, 256, 1024}) {
    const int n = mb >= 256 ? 2 : 16;
    std::vector<void*> p(n);
    const double t0 = now();
    for (int i = 0; i < n; ++i) hipHostMalloc(&p[i], mb << 20, hipHostMallocDefault);
    const double t1 = now();
    for (int i = 0; i < n; ++i) hipHostFree(p[i]);
    const double t2 = now();
    printf("hipHostMalloc 1 thread, %4zu MB blocks: %.2f GB/s (%.2f ms per block); hipHostFree %.2f GB/s\n", mb,
           n * mb / 1024.0 / (t1 - t0), 1e3 * (t1 - t0) / n, n * mb / 1024.0 / (t2 - t1));
  }
  // 2. T threads, 14 MB blocks
  for (int T : {2, 4, 8, 16}) {
    const int per = 16;
    std::vector<std::vector<void*>> p(T, std::vector<void*>(per));
    const double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        hipSetDevice(0);
        for (int i = 0; i < per; ++i) hipHostMalloc(&p[t][i], 14u << 20, hipHostMallocDefault);
      });
    for (auto& x : th) x.join();
    const double t1 = now();
    for (auto& v : p)
      for (void* q : v) hipHostFree(q);
    printf("hipHostMalloc %2d threads x %d blocks of 14 MB: %.2f GB/s aggregate\n", T, per, T * per * 14 / 1024.0 / (t1 - t0));
  }
  // 3. hipHostRegister of touched malloc memory
  {
    const size_t sz = 256u << 20;
    char* m = (char*)aligned_alloc(4096, sz);
    memset(m, 1, sz);
    const double t0 = now();
    hipHostRegister(m, sz, hipHostRegisterDefault);
    const double t1 = now();
    printf("hipHostRegister 256 MB (touched): %.2f GB/s\n", 0.25 / (t1 - t0));
    hipHostUnregister(m);
    // 4. host -> device from pageable memory
    const double t2 = now();
    for (int i = 0; i < 4; ++i) hipMemcpyAsync(d, m, sz, hipMemcpyHostToDevice, st);
    hipStreamSynchronize(st);
    const double t3 = now();
    printf("H2D from pageable memory, 4 x 256 MB: %.2f GB/s (the calls returned after %.3f s)\n", 1.0 / (t3 - t2), t3 - t2);
    void* pin = nullptr;
    hipHostMalloc(&pin, sz, hipHostMallocDefault);
    memset(pin, 1, sz);
    const double t4 = now();
    for (int i = 0; i < 4; ++i) hipMemcpyAsync(d, pin, sz, hipMemcpyHostToDevice, st);
    hipStreamSynchronize(st);
    const double t5 = now();
    printf("H2D from pinned memory,   4 x 256 MB: %.2f GB/s\n", 1.0 / (t5 - t4));
    // first-touch cost of a pinned block vs plain memory
    void* pin2 = nullptr;
    const double t6 = now();
    hipHostMalloc(&pin2, sz, hipHostMallocDefault);
    const double t7 = now();
    memset(pin2, 2, sz);
    const double t8 = now();
    char* m2 = (char*)aligned_alloc(4096, sz);
    const double t9 = now();
    memset(m2, 2, sz);
    const double t10 = now();
    printf("256 MB: hipHostMalloc %.3f s + first memset %.3f s;  plain aligned_alloc first memset %.3f s\n", t7 - t6, t8 - t7,
           t10 - t9);
  }
  return 0;
}
