// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: lane maps of A, B, D and the CBSZ / ABID broadcast controls.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma4x4_probe tools/probes/mfma4x4_probe.hip && /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CBSZ, int ABID>
__global__ void k_probe(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, CBSZ, ABID, 0);
}

template <int CBSZ, int ABID>
void run(const char* name) {
  double *da, *db, *dd;
  hipMalloc(&da, 64 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&dd, 64 * 8);
  // pairs[la][lb] -> output lane (or -1)
  std::vector<int> out(64 * 64, -1);
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double ha[64] = {0}, hb[64] = {0}, hd[64];
      ha[la] = 1.0; hb[lb] = 1.0;
      hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
      hipLaunchKernelGGL((k_probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
      hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
      int cnt = 0, first = -1;
      for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) { if (first < 0) first = l; ++cnt; }
      out[la * 64 + lb] = cnt == 0 ? -1 : (cnt == 1 ? first : 1000 + cnt * 100 + first);
    }
  printf("== %s: for each A lane la: list of (lb -> d lane[s]) ==\n", name);
  for (int la = 0; la < 64; ++la) {
    printf("la %2d:", la);
    for (int lb = 0; lb < 64; ++lb) if (out[la * 64 + lb] >= 0) printf(" %d->%d", lb, out[la * 64 + lb]);
    printf("\n");
  }
}

int main() {
  run<0, 0>("cbsz0");
  run<2, 0>("cbsz2 abid0");
  run<2, 1>("cbsz2 abid1");
  run<2, 3>("cbsz2 abid3");
  run<1, 0>("cbsz1 abid0");
  return 0;
}
