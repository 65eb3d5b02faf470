// Device helpers shared by the direct-transform kernels (af_im_to_vis.hip, af_vis_to_im.hip):
// record groups held in VGPR lanes and consumed through 64-bit DPP row_newbcast operands,
// refreshed in place by asm-issued loads that are retired with compile-time counted vmcnt waits.
#pragma once
#include <type_traits>

#include "af_common.h"

constexpr int GROUP = 16;  // doubles per record group = one DPP row

// groups of a record: 4 header doubles + ct*nc*w payload doubles
__host__ __device__ constexpr int record_groups(int ct, int nc, int w) { return (4 + ct * nc * w + GROUP - 1) / GROUP; }

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// acc += (double held by lane LANE of this lane's row of 16 in `g`) * y      [NEG: -= ]
// 64-bit DPP row_newbcast on the VOP2 form of v_fmac_f64 (full fp64 rate, measured).
// volatile: every statement that touches a record-group register stays in program order
// with the asm loads / waits below (hipcc only orders asm statements among themselves).
template <int LANE, bool NEG = false>
__device__ __forceinline__ void fmac_bcast(double &acc, double g, double y)
{
    if constexpr (NEG)
        asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(g), "v"(y), "i"(LANE));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc) : "v"(g), "v"(y), "i"(LANE));
}

// In-place refresh of one record-group register: r <- base[lane_off + OFF bytes].  Issued from
// asm because hipcc sinks a plain load down to its first use (load + vmcnt(0) back to back);
// "+v" ties the destination to the live register so no copy of a not-yet-landed value is made.
// hipcc does not count this load: group_wait<N> retires it.
template <int OFF>
__device__ __forceinline__ void group_refresh(double &r, unsigned lane_off, const double *base)
{
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "+v"(r) : "v"(lane_off), "s"(base), "i"(OFF));
}

// Wait until at most N younger vector-memory operations are outstanding; names the register it
// makes valid so that its consumers are ordered behind the wait.
template <int N>
__device__ __forceinline__ void group_wait(double &r)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "i"(N));
}

// Record-group schedule of one source iteration (all compile time).  Slots are consumed in
// increasing order: (l,m,n) at the top ("channel -1"), then channel by channel.  Group g is
// first used in channel group_first_chan(g) and refreshed for the next source right after
// channel group_last_chan(g).  At g's first use the loads issued after g's own refresh are:
// the later refreshes of the previous iteration plus the refreshes this iteration has already
// issued -- that many may stay in flight (vmcnt is in order).
__host__ __device__ constexpr int group_last_chan(int g, int nslot, int per_chan)
{
    int last = g * GROUP + GROUP - 1 < nslot - 1 ? g * GROUP + GROUP - 1 : nslot - 1;
    return last < 4 ? -1 : (last - 4) / per_chan;
}
__host__ __device__ constexpr int group_first_chan(int g, int per_chan)
{
    return g == 0 ? -1 : (g * GROUP - 4) / per_chan;
}
__host__ __device__ constexpr int group_wait_count(int g, int ng, int nslot, int per_chan)
{
    int n = 0;
    for (int h = 0; h < ng; ++h) {
        int ph = group_last_chan(h, nslot, per_chan), pg = group_last_chan(g, nslot, per_chan);
        if (ph > pg || (ph == pg && h > g)) ++n;               // previous iteration, issued after g's refresh
        if (ph < group_first_chan(g, per_chan)) ++n;            // this iteration, issued before g's first use
    }
    return n;
}

