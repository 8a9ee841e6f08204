// ref_dist_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" wrapper around the REFERENCE's DxyPerSNPfromAFs (evo_distanceToOutgroups.cpp:62-65), a free function of
// a translation unit that needs only the standard library.  The file is compiled from where it lies under
// /root/reference by including it here (-I/root/reference, the Makefile's `ref` rule); everything but the wrapper
// is hidden and garbage-collected at link time, so DistOutMain and its dependencies on process_vcf_utils.cpp (which
// needs Boost, absent here) never enter the library.  No reference source is copied and no stand-in header is used.
//
// Output: oracle/_ref/libevo_refdist.so (git-ignored, travels to the GPU box).
#include "evo_distanceToOutgroups.cpp"

extern "C" __attribute__((visibility("default")))
double ref_DxyPerSNPfromAFs(double AF1, double AF2) { return DxyPerSNPfromAFs(AF1, AF2); }
