// Small-fit translation unit: the strip-streaming fit kernel of svgp_fit.hip compiled with 256 threads per fit
// (4 waves x 256 VGPRs, two fits per CU) for M_p <= 64.  See the GAPRO_SMALL_TU block in svgp_fit.hip.
#define GAPRO_NT 256
#define GAPRO_SMALL_TU 1
#define k_svgp_fit_strip k_svgp_fit_strip256  // distinct kernel name in profiles
#include "svgp_fit.hip"
