// ref_pbs_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" wrapper around the REFERENCE's calculatePBSfromAFs (evo_PBS.cpp:69-98).  That function is `inline`
// inside the .cpp, so the translation unit is compiled from where it lies under /root/reference by including it
// here (-I/root/reference, see the Makefile's `ref` rule).  Everything but the wrapper is hidden and
// garbage-collected at link time, so PBSmain and its dependencies on the rest of the reference (which needs
// Boost, absent here) never enter the library.  No reference source is copied and no stand-in header is used.
//
// Output: oracle/_ref/libevo_refpbs.so (git-ignored, travels to the GPU box).
#include "evo_PBS.cpp"

extern "C" __attribute__((visibility("default")))
void ref_calculatePBSfromAFs(double p1, double p2, double p3, double n1, double n2, double n3, double* out) {
    std::vector<double> r = calculatePBSfromAFs(p1, p2, p3, n1, n2, n3);
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}
