// ref_abs_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" wrapper around the REFERENCE's calculateABSfromAFs (evo_ABS.cpp:58-106), built the same way as
// ref_pbs_shim.cpp: the function is `inline` inside the .cpp, so that translation unit is compiled from where it
// lies under /root/reference by including it here; everything but the wrapper is hidden and discarded at link time.
// No reference source is copied and no stand-in header is used.
//
// Output: oracle/_ref/libevo_refabs.so (git-ignored, travels to the GPU box).
#include "evo_ABS.cpp"

extern "C" __attribute__((visibility("default")))
void ref_calculateABSfromAFs(double pW, double pX, double pY, double pZ, double nW, double nX, double nY, double nZ, double* out) {
    std::vector<double> r = calculateABSfromAFs(pW, pX, pY, pZ, nW, nX, nY, nZ);
    out[0] = r[0]; out[1] = r[1];
}
