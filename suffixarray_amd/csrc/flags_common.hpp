// flags_common.hpp -- what the first flags pass of a build hands on: the query path's bucket directory and, on near-random
// text, the staged active records (sa_build.hpp: flags_kernel / flags_lite_kernel; radix_split.hpp: the local pass of the
// three-pass plan does the same work on the sub-bucket it holds in LDS).
#pragma once
#include "common.hpp"

namespace sa {

// Bucket directory of the query path: dir[bkt] = first slot whose key has top-dbits >= bkt.  Slot j owns the buckets
// (top(K[j-1]), top(K[j])]; runs of up to DIR_INLINE buckets are written by the thread that finds them, longer ones (unused
// codes of the compacted alphabet leave holes of up to 2^dbits / 8 buckets) are queued for dir_fill_kernel.
constexpr u32 DIR_INLINE = 40;   // 8 queued the (2^(dbits-25) * 8 + 1)-bucket holes behind every 5-character prefix of a 27-letter text: 531 441 atomics on one counter, +5 ms
constexpr u32 DIR_PIECE = 1u << 14;
struct DirArgs {
    u32* dir;        // [2^dbits + 1], or nullptr
    int dbits;
    uint4* gaps;     // queue of {first bucket, last bucket, value, -}
    u32* gap_count;  // zeroed by the host
    u32 gap_cap;
    DeviceStatus* dstat;
};
__device__ __forceinline__ void dir_emit(const DirArgs& d, u32 first, u32 last, u32 value) {
    if (last - first < DIR_INLINE) {
        for (u32 bkt = first; bkt <= last; ++bkt) d.dir[bkt] = value;
    } else {
        // queued in pieces of at most DIR_PIECE buckets: dir_fill_kernel gives one piece to one workgroup, and the
        // holes of a compacted alphabet reach 2^dbits / 8 buckets (one workgroup filling 2M entries took 0.2 ms)
        for (u64 f = first; f <= (u64)last; f += DIR_PIECE) {
            const u64 l = (f + DIR_PIECE - 1 < (u64)last) ? f + DIR_PIECE - 1 : (u64)last;
            const u32 slot = atomicAdd(d.gap_count, 1u);
            if (slot < d.gap_cap) d.gaps[slot] = make_uint4((u32)f, (u32)l, value, 0u);
            else __hip_atomic_store(&d.dstat->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // cannot happen: see gap_cap
        }
    }
}

// Staged active records of a near-random text (flags_lite_kernel): LITE_CAP entries of {slot, suffix, head bit} per tile
constexpr u32 LITE_CAP = 256;
struct LiteArgs {
    const u32* sa;       // suffix per slot (the sort's values)
    u32* st_pos;         // [tiles][LITE_CAP] slot
    u32* st_idx;         // [tiles][LITE_CAP] suffix
    u8* st_head;         // [tiles][LITE_CAP] 1 = first slot of its group
    u32* overflow;       // set to 1 by a tile with more than LITE_CAP active slots
};
}  // namespace sa
