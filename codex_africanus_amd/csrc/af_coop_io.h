// Record-per-lane arrays through whole cache lines (lane = one (row, chan) cell kernels).
//
// A lane's record is U x 16 bytes (a 2 x 2 complex128 cell or gain: U = 4) that its neighbours do not share an
// instruction's cache line with: read or written lane by lane, every 16-byte instruction of the wave touches 32-64
// half-used lines and the CU's address / tag path -- not HBM -- bounds the kernel (tools/microbench_gather.hip: 2.8 TB/s
// against 4.6 TB/s).  Instead U lanes move one record together (consecutive lanes = consecutive 16 bytes on the memory
// side) and the wave transposes through a private LDS region of 64 U x 16 bytes: slot 64 k + lane on the way in, an
// XOR-swizzled slot on the way out, so both directions are free of bank conflicts.  Every lane of the wave must call.
// (The calibration consumers, csrc/af_calibration.hip, carry the same scheme for their direction-stacked records.)
#pragma once
#include "af_common.h"

// rec: the lane's record index in units of U x 16 bytes (gathered records: any order; streamed cells: rec = cell number)
template <int U>
__device__ __forceinline__ void coop_gather_units(const double2 *__restrict__ src, int rec, double2 (&g)[U], double2 *lds_wave)
{
    if constexpr (U == 1) {
        g[0] = src[rec];
    } else {
        const int lane = threadIdx.x & 63;
        constexpr int CPI = 64 / U;   // records per load instruction
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int c = k * CPI + lane / U, h = lane % U;
            const int rec_c = __shfl(rec, c, 64);
            const int hs = U >= 4 ? (h ^ ((c >> 1) & (U - 1))) : h;
            lds_wave[k * 64 + lane] = src[(int64_t)rec_c * U + hs];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int slot = U >= 4 ? (j ^ ((lane >> 1) & (U - 1))) : j;
            g[j] = lds_wave[lane * U + slot];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// the wave's 64 consecutive records [wave_rec0, wave_rec0 + 64) of `out`, clipped at nrec
template <int U>
__device__ __forceinline__ void coop_store_units(double2 *__restrict__ out, int64_t wave_rec0, int64_t nrec, const double2 (&v)[U],
                                                 double2 *lds_wave)
{
    const int lane = threadIdx.x & 63;
    if constexpr (U == 1) {
        if (wave_rec0 + lane < nrec) out[wave_rec0 + lane] = v[0];
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int slot = U >= 4 ? (j ^ ((lane >> 1) & (U - 1))) : j;
            lds_wave[lane * U + slot] = v[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int CPI = 64 / U;
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int c = k * CPI + lane / U, h = lane % U;
            const int hs = U >= 4 ? (h ^ ((c >> 1) & (U - 1))) : h;
            // slot hs of record c holds element hs ^ swizzle(c) = h: consecutive lanes store consecutive 16 bytes
            const double2 x = lds_wave[c * U + hs];
            if (wave_rec0 + c < nrec) out[(wave_rec0 + c) * U + h] = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
