// commit_sort.hip — the device radix sort behind tri_commit_google: (term, document) keys with the posting's index as the value.
// Part of libtrinity_hip.so (MI355X / gfx950).  The sort itself is rocPRIM's (ROCm's own primitive library, header-only, compiled here for
// gfx950): a stable LSD radix sort over the 64 key bits.  Its own translation unit so that trinity_hip.hip does not pay for the rocPRIM
// headers at every build.  New code, no reference source.
#include <cstring> // (rocprim's texture iterator calls the host memset)
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

// tmp == nullptr: *tmp_bytes = the temporary storage the sort needs; else the sort is enqueued on `stream`.  Returns a hipError_t.
extern "C" int tri_sort_pairs_u64_u32(const unsigned long long *keys_in, unsigned long long *keys_out, const unsigned *vals_in, unsigned *vals_out, size_t n, void *tmp,
                                      size_t *tmp_bytes, hipStream_t stream) {
        return (int)rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, stream, false);
}
