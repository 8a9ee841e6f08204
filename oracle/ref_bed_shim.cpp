// ref_bed_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" wrapper around the REFERENCE's AccessibleGenome::getAccessibleBPinRegion
// (process_vcf_annotation_tools.h:360-402, :510-512), the count behind --accessibleGenomeBED
// (process_vcf_fst.cpp:593-595, :614-616; evo_distanceToOutgroups.cpp:224-229).  The header is compiled from where it
// lies under /root/reference (-I/root/reference, the Makefile's `ref` rule); the function is an inline member that needs
// only the standard library.  The BED loader next to it calls split() of process_vcf_utils.cpp (needs Boost, absent
// here), so the shim fills the feature map itself and the loader stays a restatement in the oracle.  Everything but the
// wrapper is hidden and garbage-collected at link time.  No reference source is copied and no stand-in header is used.
//
// Output: oracle/_ref/libevo_refbed.so (git-ignored, travels to the GPU box).
#include "process_vcf_annotation_tools.h"

namespace {
struct FilledAccessibleGenome : public AccessibleGenome {
    FilledAccessibleGenome(const int *starts, const int *ends, int n) {
        std::vector<std::vector<int> > f(2);
        f[0].assign(starts, starts + n);
        f[1].assign(ends, ends + n);
        BedFeatureMap["scaffold"] = f;
    }
};
}

// features of one scaffold (starts / ends in file order); "scaffold_absent" != 0 asks about a scaffold the map lacks
extern "C" __attribute__((visibility("default")))
int ref_accessible_bp(const int *starts, const int *ends, int n, int start, int end, int scaffold_absent) {
    FilledAccessibleGenome ag(starts, ends, n);
    return ag.getAccessibleBPinRegion(scaffold_absent ? "other" : "scaffold", start, end);
}
