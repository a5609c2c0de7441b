// Debug translation unit of svgp_fit.hip: the product-engine bench, the MFMA lane-map self test and the counter
// calibration streams of include/gapro_hip_debug.h, built into libgapro_hip_debug.so (never loaded by the product).
#define GAPRO_DEBUG_TU 1
#include "svgp_fit.hip"
