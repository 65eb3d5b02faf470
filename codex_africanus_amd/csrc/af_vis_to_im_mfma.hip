// vis_to_im, MFMA-accumulator path for gfx950: 4 correlations on a band whose 64-channel tiles are
// uniformly spaced.
//
// Same sum as af_vis_to_im.hip (africanus/dft/kernels.py:104-146),
//     im[s,nu,c] = sum_r [no corr of (r,nu) flagged] Re(vis[r,nu,c] e^{ip}),
// with the mapping of af_im_to_vis_mfma.hip transposed: the MFMA step contracts 4 ROWS,
//     D[source, corr] += sum_{k<4} Re Y[source, row k] * Re V[row k, corr] + Im Y * (-Im V),
// a wave owns 16 sources x a tile of CT channels (one real accumulator per (source, chan, corr): CT = 64
// uses 128 AGPRs), every lane owns one (source, row) pair of the step and computes its phasor
// (af_mfma_phasor.h).  Records per (tile, 4-row step): [(u,v,w,0) x 4 rows of the NEXT step |
// CT channels x 16 (row k, corr) x (Re V, -Im V)], flag-masked by the pack pass, copied global -> LDS one
// step ahead.  The row sum is split into row partitions (grid.z) whose partial images are added in
// partition order by v2i_reduce_kernel (deterministic, no atomics).
//
// Reference semantics kept by the pack pass: a (row, chan) with any flagged correlation contributes
// exactly nothing (zeros; a non-finite uvw row is zeroed in the header so its phasor is finite); an
// UNFLAGGED cell of a non-finite row contributes NaN to every source, as p = NaN does in the reference.
#include <stdlib.h>

#include "af_mfma_phasor.h"
#include "af_v2i_mfma.h"

namespace {

constexpr int THREADS = 256;

__host__ __device__ constexpr int v_stage_doubles(int ct) { return (2 * ct + 1) * 16; }

// per MFMA tile: quarter turns per metre at its first channel and per channel step; flags[3] &= the
// tile is an arithmetic progression within 2 ulp
__global__ void v2i_mfma_prep_freq(const double *__restrict__ freq, int64_t nchan, const int64_t *__restrict__ tile_c0,
                                   const int *__restrict__ tile_ct, int ntile, int sign, double *__restrict__ tilef,
                                   int *__restrict__ flags)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntile) return;
    const int64_t c0 = tile_c0[t];
    const int64_t nc = (nchan - c0 < tile_ct[t]) ? (nchan - c0) : tile_ct[t];
    const double f0 = freq[c0];
    const double df = (nc > 1) ? (freq[c0 + nc - 1] - f0) / (double)(nc - 1) : 0.0;
    bool uniform = isfinite(f0) && isfinite(df);
    for (int64_t j = 0; j < nc; ++j) {
        const double f = freq[c0 + j], pred = f0 + (double)j * df;
        if (!(fabs(f - pred) <= 2.0 * 2.220446049250313e-16 * fmax(fabs(f), fabs(pred)))) uniform = false;
    }
    const double s4 = 4.0 * (double)sign;
    tilef[2 * t] = s4 * f0 / AF_LIGHTSPEED;
    tilef[2 * t + 1] = s4 * df / AF_LIGHTSPEED;
    if (!uniform) atomicAnd(&flags[3], 0);
}

__global__ void v2i_mfma_fill_tiles(int64_t *c0, int *ct, int64_t nfull, int main_ct, int tail_ct, int64_t tail_c0)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nfull) { c0[t] = main_ct * t; ct[t] = main_ct; }
    if (t == nfull && tail_ct) { c0[t] = tail_c0; ct[t] = tail_ct; }
}

// Record of step `it` (4 rows) of a tile of CT channels: [(u,v,w,0) x 4 of step it + 1 | CT channels x 16 (row k, corr)
// x (Re V, -Im V)].  One thread per complex visibility (a 16-byte load and a 16-byte store; the payload's 16
// complex values per channel are contiguous, so stores coalesce fully and loads in 64-byte row segments), one
// 4-byte flag word per (row, chan); the 16 header doubles of a step are written by the threads of its first channel.
__global__ void v2i_mfma_pack(const double2 *__restrict__ vis, const unsigned char *__restrict__ vflags,
                              const double *__restrict__ uvw, const int *__restrict__ flags, int64_t nrow, int64_t nstep,
                              int64_t nchan, int64_t c0, int CT, double *__restrict__ rec, int *__restrict__ chan_any)
{
    if (flags[3] != 1) return;  // the VALU kernels own this call: their pack pass fills the region instead
    const int64_t per = v_stage_doubles(CT);
    const int per_step = CT * 16;                 // complex values of a step's payload
    const int64_t total = nstep * per_step;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t it = i / per_step;
        const int e = (int)(i - it * per_step), j = e >> 4, kn = e & 15;   // channel of the tile, (row k, corr)
        const int64_t r = 4 * it + (kn >> 2), ch = c0 + j;
        double2 out = make_double2(0.0, 0.0);
        if (r < nrow && ch < nchan) {
            const unsigned fl = *reinterpret_cast<const unsigned *>(vflags + (r * nchan + ch) * 4);
            if (fl == 0) {
                const double a = uvw[3 * r], b = uvw[3 * r + 1], c = uvw[3 * r + 2];
                const double2 x = vis[(r * nchan + ch) * 4 + (kn & 3)];
                out = make_double2(x.x, -x.y);
                if (!(isfinite(a) && isfinite(b) && isfinite(c))) out = make_double2(nan, nan);
                if ((kn & 3) == 0) chan_any[ch] = 1;  // once per unflagged (row, chan); benign race: all store 1
            }
        }
        double *step = rec + it * per;
        *reinterpret_cast<double2 *>(step + 16 + e * 2) = out;
        if (j == 0) {  // header: (u,v,w,0) of the NEXT step's row kn >> 2 ... written once, by the first channel's threads
            const int64_t rn = 4 * (it + 1) + (kn >> 2);
            double h = 0.0;
            if (rn < nrow && (kn & 3) < 3) {
                const double a = uvw[3 * rn], b = uvw[3 * rn + 1], c = uvw[3 * rn + 2];
                if (isfinite(a) && isfinite(b) && isfinite(c)) h = uvw[3 * rn + (kn & 3)];
            }
            step[kn] = h;
        }
    }
}

// grid: (ceil(nsrc/64), tiles of the launch, row partitions); block: 4 waves x 16 sources
template <int CT>
__global__ __launch_bounds__(THREADS) void v2i_mfma_kernel(
    const double *__restrict__ lmn, const double *__restrict__ uvw, const double *__restrict__ records,
    const double *__restrict__ tilef, const int *__restrict__ flags, double *__restrict__ partial, int64_t nsrc,
    int64_t nrow, int64_t nstep, int64_t steps_per_part, int64_t nchan, int64_t c0_first, int nsgroup, int npart)
{
    if (flags[3] != 1) return;
    constexpr int STAGE = v_stage_doubles(CT);
    constexpr int UNITS = STAGE / 2;
    __shared__ double smem[2 * STAGE];
    __shared__ double2 ptab[PHASOR_TABLE];   // exp(2 pi i k / 256): af_sincos.h table phasor
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = lane >> 4;  // the row of a step this lane computes the phasor of
    const int tile = blockIdx.y;
    // XCD-aware order: the hardware deals workgroups b, b + 8, ... to one XCD.  Every source group of a row partition
    // goes to ONE XCD (partition = 8 (slot / groups) + b % 8), so a partition's records are fetched into one L2 once and
    // shared by its source groups instead of being fetched by all eight (measured: 41 GB fetched for 4 GB of records)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bx = slot % nsgroup, bz = (slot / nsgroup) * 8 + xcd;
    if (bz >= npart) return;
    const int64_t c0 = c0_first + (int64_t)tile * CT;
    const double *__restrict__ rec = records + (int64_t)tile * nstep * STAGE;
    int64_t src = (int64_t)bx * 64 + wave * 16 + (lane & 15);
    if (src >= nsrc) src = nsrc - 1;
    const double l = lmn[4 * src], m = lmn[4 * src + 1], n = lmn[4 * src + 2];
    const double F0 = 64.0 * tilef[2 * tile], FD = 64.0 * tilef[2 * tile + 1];   // quarter turns -> 1/256 turns per metre (exact)
    const int64_t it0 = (int64_t)bz * steps_per_part;
    const int64_t it1 = (it0 + steps_per_part < nstep) ? it0 + steps_per_part : nstep;
    const int boff = k * 4 + (lane & 3);  // B operand (Re V, -Im V) of (row k, corr lane & 3)

    double acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = 0.0;

    mfma_stage_load<UNITS>(rec + it0 * STAGE, smem, wave, lane);
    table_phasor_init(ptab, threadIdx.x, blockDim.x);
    asm volatile("" :: "v"(l), "v"(m), "v"(n), "s"(F0), "s"(FD));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    constexpr int NGRP = CT / 8;  // channel groups of 8
    double yr[2][8], yi[2][8];
    PhasorSetup cur_, nxt_;
    {   // the partition's first step is set up in one piece from the global uvw
        const int64_t r = 4 * it0 + k;
        double a = 0.0, b = 0.0, c = 0.0;
        if (r < nrow) { a = uvw[3 * r]; b = uvw[3 * r + 1]; c = uvw[3 * r + 2]; }
        if (!(isfinite(a) && isfinite(b) && isfinite(c))) { a = 0.0; b = 0.0; c = 0.0; }
        cur_.a0 = a; cur_.a1 = b; cur_.a2 = c;
#pragma unroll
        for (int sl = 1; sl < 8; ++sl) phasor_setup_slice(cur_, sl, nullptr, l, m, n, F0, FD, ptab, yr[0], yi[0]);
    }
    nxt_ = cur_;

#pragma unroll 1
    for (int64_t it = it0; it < it1; ++it) {
        const int cur = (int)((it - it0) & 1);
        if (it + 1 < it1) mfma_stage_load<UNITS>(rec + (it + 1) * STAGE, smem + (cur ^ 1) * STAGE, wave, lane);
        const double *S = smem + cur * STAGE;  // header: (u,v,w) of the rows of step it + 1
        const double2 *B = reinterpret_cast<const double2 *>(S + 16) + boff;
        double2 bg[2][8];
        double anr = cur_.y0r, ani = cur_.y0i;
#pragma unroll
        for (int p = 0; p < 8; ++p) bg[0][p] = B[p * 16];
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            if (g + 1 < NGRP) {  // phasors of the next 8 channels
                if (((g + 1) * 8) % MFMA_ANCHOR == 0) {
                    const double tr = fma(anr, cur_.ar, -__dmul_rn(ani, cur_.ai));
                    const double ti = fma(anr, cur_.ai, __dmul_rn(ani, cur_.ar));
                    anr = tr; ani = ti;
                    phasor_first_segment(cur_, anr, ani, yr[(g + 1) & 1], yi[(g + 1) & 1]);
                } else {
                    phasor_next_segment(cur_, yr[(g + 1) & 1], yi[(g + 1) & 1], yr[g & 1], yi[g & 1]);
                }
            }
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
                if (sl * NGRP / 8 == g) phasor_setup_slice(nxt_, sl, S + 4 * k, l, m, n, F0, FD, ptab, yr[0], yi[0]);
            __builtin_amdgcn_sched_barrier(0);
            // two MFMAs per channel on ONE accumulator: all Re products of the group first, then all
            // Im products, so that dependent MFMAs are 8 instructions apart
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                acc[g * 8 + jj] = __builtin_amdgcn_mfma_f64_4x4x4f64(yr[g & 1][jj], bg[g & 1][jj].x, acc[g * 8 + jj], 0, 0, 0);
                if (g + 1 < NGRP) {
                    bg[(g + 1) & 1][jj] = B[((g + 1) * 8 + jj) * 16];
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                acc[g * 8 + jj] = __builtin_amdgcn_mfma_f64_4x4x4f64(yi[g & 1][jj], bg[g & 1][jj].y, acc[g * 8 + jj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur_ = nxt_;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // D lane = 16 i + 4 b + corr holds source 4 b + i
    const int osrc = 4 * ((lane >> 2) & 3) + (lane >> 4), ocorr = lane & 3;
    const int64_t s = (int64_t)bx * 64 + wave * 16 + osrc;
    if (s >= nsrc) return;
    const int nvalid = (int)((nchan - c0 < CT) ? (nchan - c0) : CT);
    double *__restrict__ o = partial + (((int64_t)bz * nsrc + s) * nchan + c0) * 4 + ocorr;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        if (j < nvalid) o[j * 4] = acc[j];
    }
}

struct Plan {
    int ct;              // width of the main tiles (v2i_main_ct)
    int64_t nfull;
    int tail_ct;
    int64_t tail_c0, nstep;
    size_t tilef_off, rec_off, tail_rec_off, total;
};

// Width of the band's main tiles.  32 channels (64 AGPRs + 180 VGPRs) put two waves on a SIMD where 64 channels (316
// registers) leave one, which bought im_to_vis 4 % (af_im_to_vis_mfma.hip main_ct); measured here, same box: 23.1 ms
// against 22.7 ms with 64 -- no gain, twice the partial-image blocks -- so 64 stays.  AFHIP_V2I_MFMA_CT=32 for A/B.
int v2i_main_ct()
{
    static const int env = getenv("AFHIP_V2I_MFMA_CT") ? atoi(getenv("AFHIP_V2I_MFMA_CT")) : 0;
    return env == 32 ? 32 : 64;
}

Plan make_plan(int64_t nrow, int64_t nchan)
{
    Plan p;
    const int ct = v2i_main_ct(), half = ct / 2;
    const int64_t rem = nchan % ct;
    p.ct = ct;
    p.nstep = af_cdiv(nrow > 0 ? nrow : 1, 4);
    p.nfull = nchan / ct + (rem > half ? 1 : 0);
    p.tail_ct = (rem == 0 || rem > half) ? 0 : ((ct == 64 && rem > 16) ? 32 : 16);
    p.tail_c0 = (nchan / ct) * ct;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = af_align_up(o + bytes, 256); return at; };
    p.tilef_off = take((size_t)(p.nfull + 1) * 2 * sizeof(double) + (size_t)(p.nfull + 1) * (sizeof(int64_t) + sizeof(int)));
    p.rec_off = take((size_t)p.nfull * p.nstep * v_stage_doubles(ct) * sizeof(double));
    p.tail_rec_off = take((size_t)(p.tail_ct ? p.nstep * v_stage_doubles(p.tail_ct) : 0) * sizeof(double));
    p.total = o;
    return p;
}

template <int CT>
int run_tiles(const double2 *vis, const unsigned char *vflags, const double *uvw, const double *lmn, const int *flags,
              int *chan_any, const double *tilef, double *partial, int64_t nsrc, int64_t nrow, int64_t nstep,
              int64_t nchan, int64_t c0, int64_t ntile, int64_t npart, int64_t steps_per_part, double *rec, bool prof,
              hipStream_t st)
{
    for (int64_t t = 0; t < ntile; ++t) {
        int64_t blocks = af_cdiv(nstep * CT * 16, 256);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(v2i_mfma_pack, dim3((unsigned)blocks), dim3(256), 0, st, vis, vflags, uvw, flags, nrow, nstep,
                           nchan, c0 + t * CT, CT, rec + t * nstep * v_stage_doubles(CT), chan_any);
        AF_LAUNCH_CHECK();
    }
    if (prof) af_prof_begin(st);
    const int64_t nsgroup = af_cdiv(nsrc, 64), part8 = af_cdiv(npart, 8) * 8;
    AF_REQUIRE(nsgroup * part8 < (1LL << 31), "vis_to_im: %lld source groups x %lld row partitions", (long long)nsgroup,
               (long long)part8);
    hipLaunchKernelGGL((v2i_mfma_kernel<CT>), dim3((unsigned)(nsgroup * part8), (unsigned)ntile), dim3(THREADS), 0, st, lmn,
                       uvw, rec, tilef, flags, partial, nsrc, nrow, nstep, steps_per_part, nchan, c0, (int)nsgroup,
                       (int)npart);
    if (prof) af_prof_end(st);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

}  // namespace

bool af_v2i_mfma_eligible(int64_t nchan, int64_t ncorr)
{
    return ncorr == 4 && nchan >= 14 && nchan / 32 + 1 <= 65535;
}

size_t af_v2i_mfma_workspace_bytes(int64_t nrow, int64_t nchan)
{
    return make_plan(nrow, nchan).total;
}

int af_v2i_mfma_run(const double *vis, const unsigned char *vflags, const double *uvw, const double *frequency,
                    const double *lmn, int *flags, int *chan_any, int sign, double *partial, int64_t nsrc,
                    int64_t nrow, int64_t nchan, int64_t npart, int64_t rows_per_part, int force_uniform,
                    void *workspace, hipStream_t st)
{
    const Plan p = make_plan(nrow, nchan);
    char *ws = static_cast<char *>(workspace);
    const int64_t ntile = p.nfull + (p.tail_ct ? 1 : 0);
    // tile tables: first channel and width of every MFMA tile
    double *tilef = reinterpret_cast<double *>(ws + p.tilef_off);
    int64_t *tile_c0 = reinterpret_cast<int64_t *>(tilef + 2 * (p.nfull + 1));
    int *tile_ct = reinterpret_cast<int *>(tile_c0 + (p.nfull + 1));
    hipLaunchKernelGGL(v2i_mfma_fill_tiles, dim3((unsigned)af_cdiv(ntile, 64)), dim3(64), 0, st, tile_c0, tile_ct, p.nfull,
                       p.ct, p.tail_ct, p.tail_c0);
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(v2i_mfma_prep_freq, dim3((unsigned)af_cdiv(ntile, 64)), dim3(64), 0, st, frequency, nchan, tile_c0,
                       tile_ct, (int)ntile, sign, tilef, flags);
    AF_LAUNCH_CHECK();
    if (force_uniform) AF_HIP(hipMemsetAsync(flags + 3, 1, 1, st));  // AF_DFT_RECURRENCE: caller asserts uniform spacing
    const int64_t steps_per_part = rows_per_part / 4;
    const double2 *v2 = reinterpret_cast<const double2 *>(vis);
    int rc = AF_OK;
    if (p.nfull > 0 && p.ct == 32)
        rc = run_tiles<32>(v2, vflags, uvw, lmn, flags, chan_any, tilef, partial, nsrc, nrow, p.nstep, nchan, 0, p.nfull,
                           npart, steps_per_part, reinterpret_cast<double *>(ws + p.rec_off), true, st);
    else if (p.nfull > 0)
        rc = run_tiles<64>(v2, vflags, uvw, lmn, flags, chan_any, tilef, partial, nsrc, nrow, p.nstep, nchan, 0, p.nfull,
                           npart, steps_per_part, reinterpret_cast<double *>(ws + p.rec_off), true, st);
    if (rc != AF_OK) return rc;
    double *trec = reinterpret_cast<double *>(ws + p.tail_rec_off);
    if (p.tail_ct == 32)
        rc = run_tiles<32>(v2, vflags, uvw, lmn, flags, chan_any, tilef + 2 * p.nfull, partial, nsrc, nrow, p.nstep, nchan,
                           p.tail_c0, 1, npart, steps_per_part, trec, p.nfull == 0, st);
    else if (p.tail_ct == 16)
        rc = run_tiles<16>(v2, vflags, uvw, lmn, flags, chan_any, tilef + 2 * p.nfull, partial, nsrc, nrow, p.nstep, nchan,
                           p.tail_c0, 1, npart, steps_per_part, trec, p.nfull == 0, st);
    return rc;
}
