// Fused RIME predict with per-antenna beam-cube DDEs for ANTENNA-DECOMPOSABLE uvw, as one complex GEMM per
// (timestep, channel) on the matrix cores.
//
// The reference's chain  phase_delay -> einsum -> beam_cube_dde -> predict_vis
// (africanus/rime/examples/predict.py:107-134,404-472,525; africanus/rime/phase.py:28-61;
// africanus/rime/fast_beam_cubes.py:57-240; africanus/rime/predict.py:199-212) evaluates
//
//     V_pq(t,nu) = sum_s  E_p(s,t,nu) . ( K_pq(s,nu) X_s(nu) ) . E_q(s,t,nu)^H ,   K_pq = exp(i C nu (l,m,n).uvw_pq).
//
// In every real Measurement Set the baseline coordinates are differences of per-antenna coordinates,
// uvw_pq = uvw_p - uvw_q (per timestep); then the phasor factorises, K_pq = k_p conj(k_q) with
// k_a = exp(i C nu (l,m,n).uvw_a), and with A_a = k_a E_a (2 x 2 per (antenna, source)), G_a = A_a X_s:
//
//     V_pq = sum_s G_ps A_qs^H        i.e.   M = G H^H ,   G, H : (2 nant) x (2 nsrc) complex,
//
// a dense complex matrix product per (timestep, channel): M = N = 2 nant (128 at 64 antennas), K = 2 nsrc -- the case
// the north star reserves the matrix cores for ("MFMA used only if the source x chan accumulation is reformulated as a
// genuine dense GEMM").  Per (row, chan, source) that is 64 flop (8 complex MACs) on the needed half of M, against the
// ~126 flop the lane-per-row kernel executes (phasor per (row, source), M = G E^H, acc += K M), and the phasor is
// evaluated per (antenna, source) instead of per (row, source): 31.5 x fewer at 64 antennas.
//
// One workgroup (8 waves) owns one (timestep, channel):
//   waves 4-7 (sampling)  per batch of ST sources: voxel geometry and antenna phasor of one (source, antenna) term per
//             lane, then four sampling rounds (four lanes of a quad = the four correlations of one term: the sampler of
//             af_fused_predict.hip, bilinear on the channel's pre-interpolated beam plane), E [. R], A = k E, G = A X
//             -> LDS, as MFMA operand panels  H[s][k][col]  (k = re0, re1, im0, im1 of A's row; col = 2 antenna + row)
//             and  G[s][k][row]  (k = re0, re1, im0, im1, -re0, -re1), double-buffered, one barrier per batch;
//   waves 0-3 (matrix)    the upper block triangle of M in 16 x 16 tiles (8 antennas x 8 antennas x 2 x 2), tile i on
//             wave i % 4, real and imaginary accumulators; per source and tile two v_mfma_f64_16x16x4:
//                 Cr += [Gr0 Gr1 Gi0 Gi1] . [Hr0 Hr1 Hi0 Hi1]^T      Ci += [Gi0 Gi1 -Gr0 -Gr1] . [Hr0 Hr1 Hi0 Hi1]^T
//             (Re and Im of sum_k G_k conj(H_k)); epilogue: tile element (p,i,q,j) -> row(t,p,q) of the output through
//             the plan's row map; an off-diagonal tile also serves row(t,q,p) with the conjugate transpose -- right for
//             HERMITIAN brightness matrices X_s only ((A_p X A_q^H)^H = A_q X^H A_p^H): callers send anything else to
//             af_fused_predict_c128 (include/afhip.h; rime/fused.py::_hermitian).
// fp64 MFMA and fp64 VALU share one pipe on gfx950, so the matrix cores buy no rate; what the
// formulation buys is the flop count (36 of the 64 tiles of M at 64 antennas: 73 flop per (row, chan, source) instead of
// ~126) and a pure-FMA instruction stream.
//
// Rows that are not antenna-decomposable (BASELINE's random uvw), Gaussian shapes (they depend on the baseline) and
// more than 64 antennas stay on af_fused_predict_c128.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "af_fused_device.h"

namespace {

constexpr int COL_PAD = 16;          // doubles of padding per operand plane (planes of one source on different banks)

typedef double v4d __attribute__((ext_vector_type(4)));

// Profiling build only (make HOOKS=1): shader-clock phase timers of the GEMM kernel, accumulated over every 16th
// workgroup -- [0] matrix wave 0: cycles inside the MFMA loops, [1] its cycles waiting at the batch barriers, [2] sampling
// wave 8: geometry + gather issue, [3] waiting for the gathers (an explicit vmcnt(0) is inserted for the measurement),
// [4] the four rounds' arithmetic + panel writes, [5] its barrier waits, [6] workgroups sampled, [7] batches.
#ifdef AFHIP_STAGE_HOOKS
__device__ unsigned long long g_gemm_prof[8];
#define AF_PROF_NOW() ((unsigned long long)__builtin_readcyclecounter())
#define AF_PROF_ON 1
// Timing perturbation (profiling build only): with a non-zero seed (af_debug_gemm_jitter) every wave
// sleeps a pseudo-random 0 .. ~8000 cycles -- up to a whole batch -- at about one in three of the sites around the batch
// barriers (matrix waves: behind the barrier and behind their burst; sampling waves: before / behind the barrier and
// between the rounds' arithmetic and the next geometry).  The panel-buffer protocol must not care: results are required
// bit-equal to the unperturbed kernel's in DIAG, RECT and STRADDLE instantiations (tools/stress_gemm_jitter.py).
__device__ unsigned g_gemm_jitter;
__device__ __forceinline__ void af_jitter(unsigned salt)
{
    const unsigned seed = g_gemm_jitter;
    if (seed == 0) return;
    unsigned h = seed ^ (blockIdx.x * 0x9E3779B1u) ^ (blockIdx.y * 0x85EBCA77u) ^ (blockIdx.z * 0xC2B2AE3Du) ^
                 ((salt + 64u * (threadIdx.x >> 6)) * 0x27D4EB2Fu);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
    if (h % 3u != 0) return;
    const int n = (h >> 8) & 127;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
#else
#define AF_PROF_NOW() 0ull
#define AF_PROF_ON 0
__device__ __forceinline__ void af_jitter(unsigned) {}
#endif

// ------------------------------------------------------------------------------------------------------------------
// The complex product as THREE real matrix products (the 3M / Karatsuba form): with Gs = Gr + Gi and Hd = Hr - Hi,
//     P1 = Gr Hr^T,  P2 = Gi Hi^T,  P3 = Gs Hd^T      ->      Re M = P1 + P2,   Im M = P3 - P1 + P2,
// 3 MFMAs per tile and source PAIR (K = 4 = 2 sources x 2 columns of the Jones row) where the direct form needs 4: a
// quarter of the matrix-core time for one more accumulator per tile (P1, P2, P3: 12 doubles per lane and tile) and two
// more operand planes per source (Gs, Hd).  (The direct four-product kernel of round 4's first half, 8 waves, was
// removed in round 5: 1.33 x the MFMA time and the only GEMM kernel that spilled.)  Rounding: every P is a sum of products of the
// same magnitudes as the direct form's, so the error bound is the direct form's times a small constant.
// 12 waves: waves 0-7 matrix (tile i on wave i % 8: 5 or 4 tiles at 64 antennas, 9 per SIMD), waves 8-11 sampling.
// Tile coordinates are compile-time per wave (switch on the wave number): every operand read is one lane base plus an
// immediate offset.
constexpr int G3_THREADS = 768, G3_MATRIX = 512, G3_SAMPLERS = 256, G3_PLANES = 6;

// ---- super-tiles (round 5: 65 .. 256 antennas; round 6: .. 512) ---------------------------------------------------------------------
// A workgroup owns one (timestep, channel, SUPER-TILE of M): its eight matrix waves hold at most 5 tiles each (12
// accumulator doubles per tile and lane: 168 registers at three waves per SIMD), i.e. <= 40 tiles of 8 x 8 antennas.
//   DIAG<nb>   block rows = block columns = nb consecutive 8-antenna blocks: the upper block triangle, nb (nb + 1) / 2
//              tiles; every sampled (source, antenna) term serves as a row (G planes) and as a column (H planes).
//              nant <= 64 is ONE such super-tile (round 4's kernel).
//   RECT       8 block rows x 4 block columns off the diagonal (32 tiles, matrix wave w = block row w; fewer column blocks
//              at the end of the array: those tiles are skipped): the 64 row antennas are sampled for their G planes, the
//              <= 32 column antennas for their H planes -- 96 terms per source.
// More than 64 antennas: the ceil(nant / 8) blocks are cut into super-blocks of 8 (the last one shorter); every
// super-block is a DIAG, every pair (i < j) two RECTs (rows = super-block i, columns = the halves of super-block j).
// 128 antennas: 2 x 64 + 2 x 96 = 320 sampled terms per source for 8128 baselines (64 antennas: 64 for 2016): sampling
// per baseline costs 1.24 x what it does at 64 antennas, and far less than the lane-per-row kernel's arithmetic.
// (Round 5's first tiling -- super-blocks of 4 .. 6 blocks, square RECTs of a x a tiles -- sampled 400 terms at 128
// antennas: 160.6 ms for 1e6 rows x 64 channels x 1000 sources against 206.3 for the lane-per-row kernel.)
//   FOURM      (round 6) the complex product in its direct four-product form on TWO accumulators per tile (8 doubles instead of
//              the 3M form's 12): Re += Gr Hr + Gi Hi, Im += Gi Hr + Gr (-Hi), with H planes re, im, -im and G planes re, im.
//              A matrix wave then holds EIGHT tiles in the registers that hold five in the 3M form, so a RECT super-tile can
//              be 8 x 8 blocks: 128 sampled terms per source for 64 tiles (2.0 per tile; RECT 8 x 4: 96 for 32 = 3.0).  A
//              third more matrix instructions per tile -- affordable exactly where it is used: a batch lasts as long as its
//              sampling waves' serial chains (6700 cycles per 256 terms), the matrix waves' burst hides inside it.
template <bool RECT, int NBR, int NBC, int ST, bool FOURM = false>
struct Geo {
    static constexpr int NAR = 8 * NBR, NAC = 8 * NBC;                      // antennas of the G (row) / H (column) panels
    static constexpr int CSG = 2 * NAR + COL_PAD, CSH = 2 * NAC + COL_PAD;  // doubles per operand plane
    static constexpr int HP = G3_PLANES, GP = FOURM ? 4 : G3_PLANES;        // planes of a source's H / G panels
    static constexpr int SRC = HP * CSH + GP * CSG;                         // one source: H planes, then G planes
    static constexpr int BUF = ST * SRC;                                    // one batch
    static constexpr int TPS = RECT ? NAR + NAC : NAR;                      // sampled terms (antenna slots) per source
    static constexpr int BT = ST * TPS;                                     // terms per batch
    // the 256 sampling lanes take 256 terms a super-round.  A batch that is not a whole number of super-rounds either
    // pads its last one (DIAG of small arrays: lanes redo earlier terms) or -- RECT: 4 x 96 = 384 terms -- lets
    // the super-rounds run on across the batch boundary (a flat term stream), which needs a third panel buffer: while
    // the matrix waves read batch k - 1 the samplers finish batch k and are already writing into batch k + 1.
    static constexpr bool STRADDLE = RECT && BT % G3_SAMPLERS != 0;
    static constexpr int DEPTH = STRADDLE ? 3 : 2;
    static constexpr int SRB = (BT + G3_SAMPLERS - 1) / G3_SAMPLERS;        // super-rounds per batch (not STRADDLE)
    static constexpr int NTILE = RECT ? NBR * NBC : NBR * (NBR + 1) / 2;
    static constexpr size_t lds_bytes()
    {
        return (size_t)DEPTH * BUF * sizeof(double) + (size_t)TPS * (6 + 4) * sizeof(double) + (size_t)TPS * 4 * sizeof(double2) +
               PH_TABLE * sizeof(double2);
    }
};

struct SuperTile {
    int row_ant0, col_ant0;   // first antenna of the row / column super-block
    int nc_act;               // RECT: column blocks actually present (1 .. NBC)
    int pad_;
};
struct SuperTileList {
    SuperTile e[16];
};

// Tiles of the upper block triangle in ROW-PAIR order -- row r (nb - r tiles), then row nb - 1 - r (r + 1 tiles), r = 0,
// 1, ... -- cut into 8 contiguous chunks, one per matrix wave (4 or 5 tiles at nb = 8): consecutive tiles of a wave share
// their block row, i.e. their A operand (G planes), which is then read from LDS once per row and source pair instead of
// once per tile (a third fewer operand reads than the round-robin assignment of round 4; no time gained: 96.7 vs 96.8 ms).
constexpr int pair_tile(int nb, int idx, bool col)
{
    for (int r = 0; r < (nb + 1) / 2; ++r) {
        if (idx < nb - r) return col ? r + idx : r;
        idx -= nb - r;
        const int r2 = nb - 1 - r;
        if (r2 != r) {
            if (idx < nb - r2) return col ? r2 + idx : r2;
            idx -= nb - r2;
        }
    }
    return 0;
}
template <bool RECT, int NBR, int NBC> constexpr int tile_row(int idx) { return RECT ? idx / NBC : pair_tile(NBR, idx, false); }
template <bool RECT, int NBR, int NBC> constexpr int tile_col(int idx) { return RECT ? idx % NBC : pair_tile(NBR, idx, true); }
constexpr int chunk_lo(int ntile, int c) { return c * ntile / 8; }

template <typename F, int... Js>
__device__ __forceinline__ void for_each_const(F &&fn, std::integer_sequence<int, Js...>)
{
    (fn(std::integral_constant<int, Js>{}), ...);
}

// antenna (inside its panel of 4 q4 antennas) of operand slot `slot`: the samplers' conflict-free slot order.  q4 is a
// compile-time constant except for the column panel of a RECT super-tile with fewer than NBC column blocks: its 8 nc_act
// antennas must fill the FIRST nc_act slot blocks (the tiles of the others are skipped).
__device__ __forceinline__ int slot_antenna(int slot, int q4) { return (slot % q4) * 4 + slot / q4; }
__device__ __forceinline__ int antenna_slot(int ant, int q4) { return (ant & 3) * q4 + (ant >> 2); }

// (Round 5, measured and removed: every 16 x 16 x 4 step as FOUR v_mfma_f64_4x4x4_4b on row-rotated A operands, so that the
// fp64 pipe is handed back to the SIMD's sampling wave every 16 cycles instead of every 64.  With the rotations free -- a
// deliberately wrong kernel that fed all four the same operand -- 96.7 -> 90.0 ms; with them paid for, never better than the
// plain form: DPP row_ror copies per tile 108.8 ms, per block row 98.3, the rotated operand read from LDS in rotated lane
// order 106.0.  CBSZ / ABID do not broadcast for f64: tools/probe/probe_mfma_f64_bcast.hip.  Dealing the chunks so that
// the two matrix waves of every SIMD carry 9 tiles together instead of 8 / 10 (waves w, w + 4, w + 8 share a SIMD:
// tools/probe/probe_wave_simd.hip) was slower, 102.2 vs 96.5 ms on one box.)
template <bool RECT, int NBR, int NBC, int ST, int W>
__device__ __forceinline__ void matrix_wave3(const double *__restrict__ ldsd, int nbatch, int only_stage, int lane,
                                             const int32_t *__restrict__ rm, int nap, const SuperTile tile,
                                             int64_t nchan, int64_t f, double2 *__restrict__ out, int burst_delay)
{
    using G = Geo<RECT, NBR, NBC, ST>;
    constexpr int CSG = G::CSG, CSH = G::CSH, SRC = G::SRC, BUF = G::BUF;
    constexpr int T0 = chunk_lo(G::NTILE, W), CNT = chunk_lo(G::NTILE, W + 1) - T0;
    constexpr int CN = CNT > 0 ? CNT : 1;
    const int kq = lane >> 4, c16 = lane & 15;
    // K index kq = 2 (source of the pair) + (column of the Jones row): this lane's element of every operand plane
    const int offB = (kq >> 1) * SRC + (kq & 1) * CSH + c16;
    const int offA = (kq >> 1) * SRC + G3_PLANES * CSH + (kq & 1) * CSG + c16;
    v4d p1[CN], p2[CN], p3[CN];
#pragma unroll
    for (int j = 0; j < CN; ++j) p1[j] = p2[j] = p3[j] = (v4d){0.0, 0.0, 0.0, 0.0};
    int buf = 0;
    [[maybe_unused]] unsigned long long prof_mfma = 0, prof_bar = 0, prof_t = AF_PROF_NOW();
    for (int b = 0; b < nbatch; ++b) {
        af_jitter(4u * b + 1u);
        __syncthreads();
        af_jitter(4u * b + 2u);
        if (AF_PROF_ON) { const unsigned long long n = AF_PROF_NOW(); prof_bar += n - prof_t; prof_t = n; }
        // (the SIMD's sampling wave runs order B: its rounds' arithmetic comes first after the barrier -- let it have the
        // pipe, start the burst later)
        for (int d = burst_delay; d > 0; d -= 16) __builtin_amdgcn_s_sleep(16);
        const double *P = ldsd + buf * BUF;
        buf = buf + 1 == G::DEPTH ? 0 : buf + 1;
        if (only_stage == 1 || CNT == 0) continue;
#pragma unroll
        for (int s2 = 0; s2 < ST / 2; ++s2) {
            const double *S = P + 2 * s2 * SRC;
            for_each_const([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int pb = tile_row<RECT, NBR, NBC>(T0 + j), qb = tile_col<RECT, NBR, NBC>(T0 + j);
                if (RECT && qb >= tile.nc_act) return;           // block-uniform: the column super-block is one block short
                const double *A = S + offA + pb * 16, *B = S + offB + qb * 16;
                p1[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0], B[0], p1[j], 0, 0, 0);
                p2[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[2 * CSG], B[2 * CSH], p2[j], 0, 0, 0);
                p3[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[4 * CSG], B[4 * CSH], p3[j], 0, 0, 0);
            }, std::make_integer_sequence<int, CNT>{});
        }
        if (AF_PROF_ON) { const unsigned long long n = AF_PROF_NOW(); prof_mfma += n - prof_t; prof_t = n; }
    }
#ifdef AFHIP_STAGE_HOOKS
    if (W == 0 && lane == 0 && (blockIdx.x & 15) == 0 && blockIdx.y == 0) {
        atomicAdd(&g_gemm_prof[0], prof_mfma); atomicAdd(&g_gemm_prof[1], prof_bar);
        atomicAdd(&g_gemm_prof[6], 1ull); atomicAdd(&g_gemm_prof[7], (unsigned long long)nbatch);
    }
#endif
    if (only_stage == 1) return;
    for_each_const([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int pb = tile_row<RECT, NBR, NBC>(T0 + j), qb = tile_col<RECT, NBR, NBC>(T0 + j);
        if (RECT && qb >= tile.nc_act) return;
        // operand rows / columns are antenna SLOTS (the samplers' conflict-free order): slot -> antenna
        const int jj = c16 & 1;
        const int q = tile.col_ant0 + slot_antenna(qb * 8 + (c16 >> 1), RECT ? 2 * tile.nc_act : G::NAC / 4);
        int r1[4], r2[4];                 // the tile's row-map entries first (independent loads), then the stores
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int p = tile.row_ant0 + slot_antenna(pb * 8 + ((kq + 4 * reg) >> 1), G::NAR / 4);
            r1[reg] = rm[p * nap + q];          // (< 2^16: 32-bit index arithmetic)
            r2[reg] = (RECT || pb != qb) ? rm[q * nap + p] : -1;   // the same antennas the other way round: V_qp = V_pq^H
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ii = (kq + 4 * reg) & 1;
            const double re = p1[j][reg] + p2[j][reg], im = (p3[j][reg] - p1[j][reg]) + p2[j][reg];
            if (r1[reg] >= 0) out[((int64_t)r1[reg] * nchan + f) * 4 + ii * 2 + jj] = make_double2(re, im);
            if (r2[reg] >= 0) out[((int64_t)r2[reg] * nchan + f) * 4 + jj * 2 + ii] = make_double2(re, -im);
        }
    }, std::make_integer_sequence<int, CNT>{});
}

// The four-product form's matrix wave (FOURM, RECT only: wave W = block row W, NBC tiles): two accumulators per tile.
template <int NBR, int NBC, int ST, int W>
__device__ __forceinline__ void matrix_wave4(const double *__restrict__ ldsd, int nbatch, int only_stage, int lane,
                                             const int32_t *__restrict__ rm, int nap, const SuperTile tile,
                                             int64_t nchan, int64_t f, double2 *__restrict__ out)
{
    using G = Geo<true, NBR, NBC, ST, true>;
    constexpr int CSG = G::CSG, CSH = G::CSH, SRC = G::SRC, BUF = G::BUF;
    static_assert(NBR == 8, "one block row per matrix wave");
    const int kq = lane >> 4, c16 = lane & 15;
    const int offB = (kq >> 1) * SRC + (kq & 1) * CSH + c16;
    const int offA = (kq >> 1) * SRC + G::HP * CSH + (kq & 1) * CSG + c16 + W * 16;
    v4d cr[NBC], ci[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) cr[j] = ci[j] = (v4d){0.0, 0.0, 0.0, 0.0};
    int buf = 0;
    for (int b = 0; b < nbatch; ++b) {
        af_jitter(4u * b + 1u);
        __syncthreads();
        af_jitter(4u * b + 2u);
        const double *P = ldsd + buf * BUF;
        buf = buf + 1 == G::DEPTH ? 0 : buf + 1;
        if (only_stage == 1) continue;
#pragma unroll
        for (int s2 = 0; s2 < ST / 2; ++s2) {
            const double *A = P + 2 * s2 * SRC + offA, *Bp = P + 2 * s2 * SRC + offB;
            const double gr = A[0], gi = A[2 * CSG];
            for_each_const([&](auto jc) {
                constexpr int qb = decltype(jc)::value;
                if (qb >= tile.nc_act) return;                   // block-uniform: the column super-block is short
                const double *B = Bp + qb * 16;
                const double hr = B[0], hi = B[2 * CSH], nhi = B[4 * CSH];
                cr[qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(gr, hr, cr[qb], 0, 0, 0);
                ci[qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(gi, hr, ci[qb], 0, 0, 0);
                cr[qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(gi, hi, cr[qb], 0, 0, 0);
                ci[qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(gr, nhi, ci[qb], 0, 0, 0);
            }, std::make_integer_sequence<int, NBC>{});
        }
    }
    if (only_stage == 1) return;
    for_each_const([&](auto jc) {
        constexpr int qb = decltype(jc)::value;
        if (qb >= tile.nc_act) return;
        const int jj = c16 & 1;
        const int q = tile.col_ant0 + slot_antenna(qb * 8 + (c16 >> 1), 2 * tile.nc_act);
        int r1[4], r2[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int p = tile.row_ant0 + slot_antenna(W * 8 + ((kq + 4 * reg) >> 1), G::NAR / 4);
            r1[reg] = rm[p * nap + q];
            r2[reg] = rm[q * nap + p];          // the same antennas the other way round: V_qp = V_pq^H
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ii = (kq + 4 * reg) & 1;
            const double re = cr[qb][reg], im = ci[qb][reg];
            if (r1[reg] >= 0) out[((int64_t)r1[reg] * nchan + f) * 4 + ii * 2 + jj] = make_double2(re, im);
            if (r2[reg] >= 0) out[((int64_t)r2[reg] * nchan + f) * 4 + jj * 2 + ii] = make_double2(re, -im);
        }
    }, std::make_integer_sequence<int, NBC>{});
}

// grid: (nsteps, channels of the plane group, super-tiles of this shape); block 768.
template <bool FEED, bool RECT, int NBR, int NBC, int ST, bool FOURM = false>
__global__ __launch_bounds__(G3_THREADS) void fused_gemm3_kernel(
    const double *__restrict__ ant_uvw, const int32_t *__restrict__ rowmap, const double *__restrict__ lmn,
    const double *__restrict__ f4, const double2 *__restrict__ brightness, const double *__restrict__ vrec,
    int64_t beam_lw, int64_t beam_mh, int64_t beam_nud, const double *__restrict__ lm_ext,
    const double *__restrict__ freq_data, const double *__restrict__ parangles, const double *__restrict__ point_errors,
    const double *__restrict__ antenna_scaling, const double2 *__restrict__ feed_rot, int nsrc, int64_t nchan,
    int64_t ntime, int nant, int nap, double2 *__restrict__ out, int only_stage, int64_t f0, int sample_prio,
    int order_b_mask, int burst_delay, const SuperTileList tiles)
{
    static_assert(ST % 2 == 0, "sources are consumed in pairs");
    static_assert(!FOURM || RECT, "the four-product form serves RECT super-tiles");
    using G = Geo<RECT, NBR, NBC, ST, FOURM>;
    constexpr int NAR = G::NAR, NAC = G::NAC, CSG = G::CSG, CSH = G::CSH, SRC = G::SRC, BUF = G::BUF, TPS = G::TPS, BT = G::BT;
    extern __shared__ double ldsd[];
    double *ldsA = ldsd + G::DEPTH * BUF;                      // six planes of per-slot constants
    double *ldsU = ldsA + 6 * TPS;                             // (u, v, w) FT per slot, three planes (+ one of padding)
    double2 *ldsR = reinterpret_cast<double2 *>(ldsU + 4 * TPS);
    double2 *ldsT = ldsR + 4 * TPS;
    const int tid = threadIdx.x;
    const int ptid = tid - G3_MATRIX;
    const int64_t f = f0 + blockIdx.y;
    const int t = blockIdx.x;
    const SuperTile tile = tiles.e[blockIdx.z];
    // antenna of sampling slot a: DIAG -- the super-block's antennas; RECT -- the column super-block's, then the row one's
    auto slot_ant = [&](int a) { return RECT ? (a < NAC ? tile.col_ant0 + a : tile.row_ant0 + a - NAC) : tile.row_ant0 + a; };
    auto slot_ok = [&](int a) {
        const int ant = slot_ant(a);
        return ant < nant && (!RECT || a >= NAC || a < 8 * tile.nc_act);
    };

    fine_table_init(ldsT, tid, G3_THREADS);
    const double FT = f4[f] * (PH_TABLE / 4.0);
    const BeamGrid<double> bg = beam_grid<double>(lm_ext, beam_lw, beam_mh, beam_nud);
    const double fscale = freq_data[3 * f + 0];
    for (int a = tid; a < TPS; a += G3_THREADS) {
        double sp = 0.0, cp = 1.0, pl = 0.0, pm = 0.0, sl_ = 1.0, sm_ = 1.0, u = 0.0, v = 0.0, w = 0.0;
        if (slot_ok(a)) {
            const int ant = slot_ant(a);
            sincos(parangles[(int64_t)t * nant + ant], &sp, &cp);
            const double *pe = point_errors + (((int64_t)t * nant + ant) * nchan + f) * 2;
            const double *as = antenna_scaling + ((int64_t)ant * nchan + f) * 2;
            pl = pe[0]; pm = pe[1]; sl_ = as[0]; sm_ = as[1];
            const double *x = ant_uvw + ((int64_t)t * nant + ant) * 3;
            u = __dmul_rn(x[0], FT); v = __dmul_rn(x[1], FT); w = __dmul_rn(x[2], FT);
        }
        // the slot's coordinate map, folded (fused_voxels_folded): one plane per coefficient (consecutive lanes =
        // consecutive slots read consecutive doubles: no bank conflicts)
        const double kl = sl_ * bg.lscale, km = sm_ * bg.mscale;
        ldsA[0 * TPS + a] = fscale * cp * kl;                               // c1l
        ldsA[1 * TPS + a] = -fscale * sp * kl;                              // c2l
        ldsA[2 * TPS + a] = (pl * cp - pm * sp) * kl - bg.lower_l * bg.lscale;   // c0l
        ldsA[3 * TPS + a] = fscale * sp * km;                               // c1m
        ldsA[4 * TPS + a] = fscale * cp * km;                               // c2m
        ldsA[5 * TPS + a] = (pl * sp + pm * cp) * km - bg.lower_m * bg.mscale;   // c0m
        ldsU[0 * TPS + a] = u; ldsU[1 * TPS + a] = v; ldsU[2 * TPS + a] = w;
    }
    if constexpr (FEED)
        for (int i = tid; i < 4 * TPS; i += G3_THREADS)
            ldsR[i] = slot_ok(i >> 2) ? feed_rot[((int64_t)t * nant + slot_ant(i >> 2)) * 4 + (i & 3)] : make_double2(0.0, 0.0);
    __syncthreads();
    const int nbatch = (nsrc + ST - 1) / ST;

    if (tid < G3_MATRIX) {
        const int32_t *rm = rowmap + (int64_t)t * nap * nap;
        const int lane = tid & 63;
        // waves w, w + 4 and the sampling wave 8 + (w & 3) share a SIMD (tools/probe/probe_wave_simd.hip)
        const int wdelay = ((order_b_mask >> ((tid >> 6) & 3)) & 1) ? burst_delay : 0;
        if constexpr (FOURM) {
            switch (tid >> 6) {
            case 0: matrix_wave4<NBR, NBC, ST, 0>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out); break;
            case 1: matrix_wave4<NBR, NBC, ST, 1>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out); break;
            case 2: matrix_wave4<NBR, NBC, ST, 2>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out); break;
            case 3: matrix_wave4<NBR, NBC, ST, 3>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out); break;
            case 4: matrix_wave4<NBR, NBC, ST, 4>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out); break;
            case 5: matrix_wave4<NBR, NBC, ST, 5>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out); break;
            case 6: matrix_wave4<NBR, NBC, ST, 6>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out); break;
            default: matrix_wave4<NBR, NBC, ST, 7>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out); break;
            }
            return;
        }
        switch (tid >> 6) {
        case 0: matrix_wave3<RECT, NBR, NBC, ST, 0>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out, wdelay); break;
        case 1: matrix_wave3<RECT, NBR, NBC, ST, 1>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out, wdelay); break;
        case 2: matrix_wave3<RECT, NBR, NBC, ST, 2>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out, wdelay); break;
        case 3: matrix_wave3<RECT, NBR, NBC, ST, 3>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out, wdelay); break;
        case 4: matrix_wave3<RECT, NBR, NBC, ST, 4>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out, wdelay); break;
        case 5: matrix_wave3<RECT, NBR, NBC, ST, 5>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out, wdelay); break;
        case 6: matrix_wave3<RECT, NBR, NBC, ST, 6>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out, wdelay); break;
        default: matrix_wave3<RECT, NBR, NBC, ST, 7>(ldsd, nbatch, only_stage, lane, rm, nap, tile, nchan, f, out, wdelay); break;
        }
        return;
    }

    // =================================== sampling waves (8-11) ======================================
    FusedGrid grid;
    {
        grid.lower_l = grid.lower_m = grid.lscale = grid.mscale = 0.0;      // folded into the slots' coefficients
        grid.lmaxf = wave_uniform(bg.lmaxf); grid.mmaxf = wave_uniform(bg.mmaxf);
        grid.lmaxi = __builtin_amdgcn_readfirstlane((int)bg.lmaxi); grid.mmaxi = __builtin_amdgcn_readfirstlane((int)bg.mmaxi);
        grid.stride_m = VREC * 8u;
        grid.stride_l = (unsigned)beam_mh * grid.stride_m;
    }
    const int e_corr = ptid & 3;
    const int ei = e_corr >> 1, ej = e_corr & 1;
    const char *plane = reinterpret_cast<const char *>(vrec + (int64_t)blockIdx.y * beam_lw * beam_mh * VREC);
    const unsigned corr_off = e_corr * 32u;
    if (sample_prio >= 0) __builtin_amdgcn_s_setprio(3);
    // This lane's term of super-round sr: (panel buffer, source slot of the batch, antenna slot, global source).  Every
    // lane owns a term and every term is written (no branch around the stores: the four rounds below stay one scheduling
    // region): lanes beyond the batch's / the stream's terms redo an earlier term (same values to the same place), padded
    // antennas and sources beyond the last write zeros.
    struct Term {
        int buf, e_sl, slot, src;
    };
    const int total_terms = nbatch * BT;
    // batch b, super-round starting at term task0 of the batch (batches of whole super-rounds, or padded ones)
    auto term_at = [&](int b, int task0) {
        Term T;
        int task = task0 + ptid;
        if (task >= BT) task %= BT;
        T.e_sl = task / TPS;
        T.slot = task - T.e_sl * TPS;
        T.buf = b & 1;
        T.src = b * ST + T.e_sl;
        return T;
    };
    // super-round sr of the flat term stream (STRADDLE)
    auto term_flat = [&](int sr) {
        Term T;
        int x = sr * G3_SAMPLERS + ptid;
        if (x >= total_terms) x = total_terms - 1 - (x - total_terms) % BT;      // the stream's tail: a term of the last batch again
        const int sg = x / TPS;                    // source slot of the padded stream
        T.slot = x - sg * TPS;
        const int bb = sg / ST;
        T.e_sl = sg - bb * ST;
        T.buf = bb % G::DEPTH;
        T.src = sg;
        return T;
    };
    // A lane's source coordinates are fetched ONE super-round ahead: a batch is a chain of dependent memory round trips
    // (coordinates -> voxel geometry -> gathers) and the first is taken off its critical path.
    struct Coords {
        double2 lm;
        double n;
    };
    auto fetch = [&](const Term &T) {
        const bool have = T.src < nsrc && slot_ok(T.slot);
        int sidx = have ? T.src : 0;
        // TPS == 64: the wave's 64 lanes are one source's slots -- its coordinates are wave-uniform: a scalar load into
        // scalar registers (six vector registers fewer in a sampler that has none to spare)
        if constexpr (TPS == 64) sidx = __builtin_amdgcn_readfirstlane(sidx);
        const double *sp = lmn + 4 * sidx;
        Coords c;
        c.lm = *reinterpret_cast<const double2 *>(sp);
        c.n = sp[2];
        return c;
    };
    Coords nxt = fetch(G::STRADDLE ? term_flat(0) : term_at(0, 0));
    [[maybe_unused]] unsigned long long prof_geo = 0, prof_wait = 0, prof_fin = 0, prof_sbar = 0, prof_st = AF_PROF_NOW();
    auto prof_mark = [&](unsigned long long &acc) {
        if (AF_PROF_ON) { const unsigned long long n = AF_PROF_NOW(); acc += n - prof_st; prof_st = n; }
    };
    // ---- one super-round = this lane's own term (geometry, antenna phasor) + four sampling ROUNDS ----------------------
    // (in round QL the four lanes of a quad take the geometry of quad lane QL and sample one correlation each)
    struct Own {                 // a lane's own term of one super-round
        FusedVoxelsC gx;
        C2 kph;
        int info, slot, src;     // info: LDS offset (doubles, panel buffer included) of the term's planes | col_term << 30 | have << 31
        double2 xw[4];           // TPS == 64: the wave's source's brightness (scalar registers)
        double2 xb0, xb1;        // otherwise: column ej of the QUAD's source's brightness (see geometry)
    };
    struct Round {               // one round's gathers (what identifies the round's term is re-broadcast when it is consumed)
        double2 v[4];
        double ab[4];
    };
    auto geometry = [&](const Term &T, const Term &Tn, Own &S) {
        const int e_sl = T.e_sl;
        int e_slot = T.slot;
        // (the slot is the same in every super-round when a batch is exactly one super-round -- 64 antennas -- and the
        // compiler then keeps the LDS addresses of the nine per-slot constants below in nine registers across the loop:
        // the sampler has none to spare (168 with the four rounds in flight), and what it spilled it reloaded behind an
        // s_waitcnt vmcnt(0) that drained the gathers.  Opaque per super-round: nine adds instead.)
        asm volatile("" : "+v"(e_slot));
        const bool have = T.src < nsrc && slot_ok(e_slot);
        // operand address of this lane's own term and what it writes: its H planes (column antenna), its G planes
        // (row antenna), or both (DIAG)
        const bool col_term = !RECT || e_slot < NAC;
        // (the slots of an absent last column block -- RECT, nc_act < NBC -- keep their own places behind the real ones)
        const int h_slot = !RECT ? antenna_slot(e_slot, NAC / 4)
                                 : (!col_term ? 0 : (e_slot < 8 * tile.nc_act ? antenna_slot(e_slot, 2 * tile.nc_act) : e_slot));
        const int h_off = T.buf * BUF + e_sl * SRC + 2 * h_slot;
        const int g_off = T.buf * BUF + e_sl * SRC + G::HP * CSH + 2 * antenna_slot(RECT ? (col_term ? 0 : e_slot - NAC) : e_slot, NAR / 4);
        S.info = (RECT ? (col_term ? h_off : g_off) : h_off) | (int)((unsigned)col_term << 30) | (int)((unsigned)have << 31);
        S.slot = e_slot;
        // TPS == 64: a wave's 64 lanes are ONE source's slots -- its 2 x 2 complex brightness comes in by ONE scalar
        // load per super-round, here (inside the sampling rounds every round waited out an SMEM round trip --
        // lgkmcnt is shared with the LDS reads --: the sampling alone ran 65.1 ms with those loads, 51.6 without)
        if constexpr (TPS == 64) {
            const int us = __builtin_amdgcn_readfirstlane(have ? T.src : 0);
            const double2 *bp = brightness + ((int64_t)us * nchan + f) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) S.xw[c] = bp[c];
        }
        S.src = have ? T.src : 0;
        if constexpr (TPS != 64) {
            // The four terms of a quad are four consecutive slots of ONE source (TPS and the super-round's first term
            // are multiples of 4), so the brightness column this lane multiplies with in each of the four rounds is the
            // same two values: loaded HERE, once per super-round and ahead of the gathers.  (Until round 6 every round
            // loaded them again inside its arithmetic -- behind the gathers in the vector-memory queue, i.e. four more
            // dependent memory round trips per super-round: the RECT super-tile's batch took 13 100 cycles where a DIAG
            // batch of 36 tiles takes 6 700.)  A padded slot (have == false) still belongs to a real source.
            const double2 *bp = brightness + ((int64_t)(T.src < nsrc ? T.src : 0) * nchan + f) * 4;
            S.xb0 = bp[ej]; S.xb1 = bp[2 + ej];
        }
        const double2 lm2 = nxt.lm;
        const double nn = nxt.n;
        nxt = fetch(Tn);
        fused_voxels_folded(grid, lm2.x, lm2.y, ldsA[0 * TPS + e_slot], ldsA[1 * TPS + e_slot], ldsA[2 * TPS + e_slot],
                            ldsA[3 * TPS + e_slot], ldsA[4 * TPS + e_slot], ldsA[5 * TPS + e_slot], S.gx);
        S.kph = table_phasor(ldsT, fma(nn, ldsU[2 * TPS + e_slot],
                                       fma(lm2.y, ldsU[1 * TPS + e_slot], __dmul_rn(lm2.x, ldsU[0 * TPS + e_slot]))));
    };
    auto issue = [&](auto lane_c, const Own &S, Round &R) {
        constexpr int QL = decltype(lane_c)::value;
        const unsigned base = (unsigned)quad_bcast<QL>((int)S.gx.base) + corr_off;
        const unsigned dl = (unsigned)quad_bcast<QL>((int)S.gx.dl), dm = (unsigned)quad_bcast<QL>((int)S.gx.dm);
        const unsigned offs[4] = {base, base + dl, base + dm, base + dl + dm};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double *r = reinterpret_cast<const double *>(plane + (size_t)offs[k]);
            R.v[k] = *reinterpret_cast<const double2 *>(r);
            R.ab[k] = r[2];
        }
    };
    auto finish = [&](auto lane_c, const Own &S, const Round &R) {
        constexpr int QL = decltype(lane_c)::value;
        const int info = quad_bcast<QL>(S.info);
        const bool r_have = info < 0;
        const double ld = quad_bcast<QL>(S.gx.ld), md = quad_bcast<QL>(S.gx.md);
        const double omld = __dsub_rn(1.0, ld), ommd = __dsub_rn(1.0, md);
        const double wt[4] = {__dmul_rn(omld, ommd), __dmul_rn(ld, ommd), __dmul_rn(omld, md), __dmul_rn(ld, md)};
        C2 kk;
        kk.re = quad_bcast<QL>(S.kph.re); kk.im = quad_bcast<QL>(S.kph.im);
        double2 e2 = beam_reduce1(R.v, R.ab, wt);
        if (!r_have) e2 = make_double2(0.0, 0.0);
        C2 e;
        e.re = e2.x; e.im = e2.y;
        if constexpr (FEED) {
            C2 E0, E1;
            E0.re = pair_bcast<0>(e.re); E0.im = pair_bcast<0>(e.im);
            E1.re = pair_bcast<1>(e.re); E1.im = pair_bcast<1>(e.im);
            const int r_slot = quad_bcast<QL>(S.slot);
            const double2 r0 = ldsR[4 * r_slot + ej], r1 = ldsR[4 * r_slot + 2 + ej];
            C2 Q0, Q1;
            Q0.re = r0.x; Q0.im = r0.y; Q1.re = r1.x; Q1.im = r1.y;
            e = cmul(E0, Q0);
            cmac(e, E1, Q1);
        }
        const C2 A = cmul(kk, e);
        C2 A0, A1, B0, B1;
        A0.re = pair_bcast<0>(A.re); A0.im = pair_bcast<0>(A.im);
        A1.re = pair_bcast<1>(A.re); A1.im = pair_bcast<1>(A.im);
        if constexpr (TPS == 64) {
            // the wave's source: its brightness matrix was loaded with the super-round's geometry
            const double2 b0 = ej ? S.xw[1] : S.xw[0], b1 = ej ? S.xw[3] : S.xw[2];
            B0.re = b0.x; B0.im = b0.y; B1.re = b1.x; B1.im = b1.y;
        } else {
            B0.re = S.xb0.x; B0.im = S.xb0.y; B1.re = S.xb1.x; B1.im = S.xb1.y;
        }
        C2 Gv = cmul(A0, B0);
        cmac(Gv, A1, B1);
        if constexpr (!RECT) {
            // DIAG: NAR == NAC, the G planes of a slot lie G3_PLANES * CSH doubles behind its H planes
            double *hs = ldsd + (info & 0x3fffffff) + ei;
            double *gs = hs + G3_PLANES * CSH;
            hs[ej * CSH] = A.re; hs[(2 + ej) * CSH] = A.im; hs[(4 + ej) * CSH] = __dsub_rn(A.re, A.im);
            gs[ej * CSG] = Gv.re; gs[(2 + ej) * CSG] = Gv.im; gs[(4 + ej) * CSG] = __dadd_rn(Gv.re, Gv.im);
        } else {
            // RECT: a column antenna's term writes its H planes, a row antenna's its G planes: one address, one
            // plane stride and three values picked per lane, the stores themselves unconditional
            const bool r_col = (info >> 30) & 1;
            double *ws = ldsd + (info & 0x3fffffff) + ei;
            const int cs = r_col ? CSH : CSG;
            const double v0 = r_col ? A.re : Gv.re, v1 = r_col ? A.im : Gv.im;
            if constexpr (!FOURM) {
                const double v2 = r_col ? __dsub_rn(A.re, A.im) : __dadd_rn(Gv.re, Gv.im);
                ws[ej * cs] = v0; ws[(2 + ej) * cs] = v1; ws[(4 + ej) * cs] = v2;
            } else {
                // four-product form: H planes re, im, -im; G planes re, im (a row antenna's term has no third plane)
                ws[ej * cs] = v0; ws[(2 + ej) * cs] = v1;
                if (r_col) ws[(4 + ej) * CSH] = -A.im;
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // position of a super-round in the kernel's sequence, either mode
    struct Pos {
        int b, task0, sr;
    };
    auto term_of = [&](const Pos &p) { return G::STRADDLE ? term_flat(p.sr) : term_at(p.b, p.task0); };
    auto next_pos = [&](const Pos &p) {
        Pos q = p;
        q.sr = p.sr + 1;
        if constexpr (!G::STRADDLE) {
            q.task0 = p.task0 + G3_SAMPLERS;
            if (q.task0 >= BT) { q.task0 = 0; q.b = p.b + 1; }
        } else {
            q.b = (q.sr * G3_SAMPLERS) / BT;         // the batch in which super-round q.sr STARTS
        }
        return q;
    };
    const int nsr = G::STRADDLE ? (total_terms + G3_SAMPLERS - 1) / G3_SAMPLERS : nbatch * G::SRB;
    auto valid = [&](const Pos &p) { return p.sr < nsr; };
    // true when super-round p completes a batch: the batch barrier follows it
    auto completes = [&](const Pos &p) {
        if constexpr (!G::STRADDLE) return p.task0 + G3_SAMPLERS >= BT;
        else return ((p.sr + 1) * G3_SAMPLERS) / BT > (p.sr * G3_SAMPLERS) / BT || p.sr + 1 == nsr;
    };
    if (only_stage == 2) {
        for (int b = 0; b < nbatch; ++b) __syncthreads();
        return;
    }
    // (three rounds in flight at most: the fourth round's gathers land in the first's registers.  The two orders keep
    // their own copies of the round / term records: declared once for both, the register allocator spilled)
    // the samplers' side of a batch barrier when gathers are in flight across it: their panel writes (LDS) must have
    // landed, the gathers (vmcnt) stay outstanding -- __syncthreads() would drain them
    auto sampler_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // Two orders of the same super-round.  Phase timers of order A beside the matrix waves (profiling build,
    // tools/gemm_phase_timers.py; shader-clock cycles per batch at 64 antennas): geometry + gather ISSUE 4700 (3100 with
    // the matrix waves idle: the L1's line rate -- the four sampling waves' 1024 missed 128-byte lines per batch --, not
    // instructions), gather wait 850, the rounds' arithmetic 1800, batch barrier 1000; a matrix wave's MFMA loop 1700 of
    // the 8300.  All four waves in order A hit the L1 together and then compute together.
    //   order A:  geometry, issue all four rounds | wait | the four rounds' arithmetic | batch barrier
    //   order B:  the rounds of the gathers issued BEFORE the previous barrier | geometry + issue of the next super-round |
    //             batch barrier (gathers in flight across it)
    // MIXED: sampling waves 8, 9 run order A, waves 10, 11 order B -- one pair's gather issue falls into the other pair's
    // arithmetic, the L1 sees two waves at a time all the time.  (All four in order B: 109 ms against 97 -- the rounds'
    // arithmetic then meets the matrix waves' burst on every SIMD.)
    // ONE loop serves both orders -- prologue: prepare super-round 0; body: consume k, [order A: batch barrier], prepare
    // k + 1, [order B: batch barrier] -- so that the rounds' and the term's registers are the same in both (two loops in one
    // kernel made the register allocator spill).
    const bool order_b = (order_b_mask >> (ptid >> 6)) & 1;
    {
        Round R0, R1, R2, R3;
        Own S;
        Pos p = {0, 0, 0};
        geometry(term_of(p), term_of(next_pos(p)), S);
        issue(I0{}, S, R0); issue(I1{}, S, R1); issue(I2{}, S, R2); issue(I3{}, S, R3);
        prof_mark(prof_geo);
        while (true) {
            finish(I0{}, S, R0); finish(I1{}, S, R1); finish(I2{}, S, R2); finish(I3{}, S, R3);
            prof_mark(prof_fin);
            // (round 5, measured and removed: an 8-wave kernel -- FOUR matrix waves, one per SIMD with nine tiles each, + the four
            // sampling waves on a 256-register budget, the sampler software-pipelined over half super-rounds (two rounds'
            // gathers always in flight while the other two are consumed) -- 99.6 ms against 87.7, same box; phase timers: its
            // sampling stage ALONE 5150 cycles per batch against ~5300: the stage is bound by the L1's miss rate, ~5 cycles per
            // 128-byte line and CU = 66 ms for the kernel's 1.04 TB of records, pipelined or not)
            // (round 5, measured and removed, same box, same bits, 87.7 ms as it stands:
            //  * a ROLLING order -- round q of super-round k + 1 issued right after round q of k is consumed, k + 1's three
            //    address words computed beforehand and the rest of its term parked in LDS: 164 registers, no spill,
            //    vmcnt(24) waits, 24-32 gathers in flight all the time -- 93.2 ms.  The rounds' arithmetic then runs beside
            //    the matrix waves' burst: a sampling wave issues in order, and every fp64 operation of its dependent chains
            //    that becomes ready while a 64-cycle MFMA holds the SIMD's pipe waits that MFMA out, priority or not.
            //  * order C -- the next term's geometry BEFORE the batch barrier (pipe free), only the gathers' issue after
            //    it -- 94.6 ms: the gathers then land ~1200 cycles earlier and the rounds' arithmetic starts inside the
            //    matrix waves' burst instead of behind it.  Order A's geometry is the delay that keeps the two apart.
            //  * s_nop gaps of 8 or 16 cycles behind every MFMA of the matrix waves (windows for the sampling wave's
            //    operations): 108 ms in order A, 93.6 in order C.
            //  * a SECOND barrier per batch behind the matrix waves' burst (the rounds' arithmetic held back until the pipe
            //    is free), gathers issued before the first barrier: 128.7 ms; after it: 103.0 (the gather issue beside the
            //    burst: 4300 cycles instead of 3100); the sixteen gather addresses computed before the batch barrier and only
            //    the 32 loads behind it: 93.2, with the second barrier 102.4.  The gather issue wants the CU to itself.
            //  The batch is the SUM of the burst (3460 pipe cycles per SIMD) and the sampling wave's own chains (~3100
            //  elapsed at half the pipe's rate), with the gathers' latency hidden under the burst: 6700 cycles = 87.5 ms.)
            // (the two conditional barrier sites below are also what keeps the consume / barrier / geometry phases in
            // separate scheduling regions: with ONE unconditional barrier here -- order A hard-wired -- hipcc merges the
            // phases into one block and the same kernel takes 107.5 ms instead of 87.7, same box; with the matrix waves'
            // burst-delay loop removed 89.0.  Measured with tools/ab_libs.sh on three builds; do not "simplify".)
            const Pos q = next_pos(p);
            const bool more = valid(q), bar = completes(p);
            af_jitter(4u * p.sr + 1u);
            if (bar && !order_b) { sampler_barrier(); prof_mark(prof_sbar); }
            af_jitter(4u * p.sr + 2u);
            if (more) {
                geometry(term_of(q), term_of(next_pos(q)), S);
                issue(I0{}, S, R0); issue(I1{}, S, R1); issue(I2{}, S, R2); issue(I3{}, S, R3);
                prof_mark(prof_geo);
            }
            if (bar && order_b) { sampler_barrier(); prof_mark(prof_sbar); }
            af_jitter(4u * p.sr + 3u);
            if (!more) break;
            p = q;
        }
    }
#ifdef AFHIP_STAGE_HOOKS
    if (ptid == 0 && (blockIdx.x & 15) == 0 && blockIdx.y == 0) {
        atomicAdd(&g_gemm_prof[2], prof_geo); atomicAdd(&g_gemm_prof[3], prof_wait);
        atomicAdd(&g_gemm_prof[4], prof_fin); atomicAdd(&g_gemm_prof[5], prof_sbar);
    }
#endif
}

}  // namespace

#ifdef AFHIP_STAGE_HOOKS
// profiling build only: seed of the timing perturbation (0 = off)
AF_EXPORT int af_debug_gemm_jitter(unsigned seed)
{
    AF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_jitter), &seed, sizeof(seed)));
    return AF_OK;
}

// profiling build only: read (and optionally reset) the GEMM kernel's phase timers
AF_EXPORT int af_debug_gemm_prof(unsigned long long *out8, int reset)
{
    if (out8) AF_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gemm_prof), 8 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        AF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), zero, sizeof(zero)));
    }
    return AF_OK;
}
#endif

// Host-side planner (HOST pointers): is uvw antenna-decomposable, and if so with which antenna coordinates?
//   Per timestep (time_index - min): the baselines of the step form a graph on the antennas; a breadth-first walk of
//   every connected component integrates uvw along a spanning tree (x_root = 0, x_q = x_p - uvw_pq), two Gauss-Seidel
//   sweeps of the normal equations (x_p <- mean of what its baselines imply) pull that towards the least-squares
//   solution, and the component's mean is removed (small coordinates -> small phase arguments).  The plan is accepted
//   when  max_rows |x_p - x_q - uvw_pq|_inf <= tol  [metres]: a residual delta changes a source's phase by at most
//   2 pi nu / c |(l, m, n)| delta  (1.8e-10 rad at tol = 1e-10 m, 1.7 GHz, 0.05 rad from the phase centre).
//   ant_uvw_host (nsteps, nant, 3) and rowmap_host (nsteps, nap, nap), nap = 8 ceil(nant / 8): row of baseline
//   (antenna1 = p, antenna2 = q) of the step, or -1.  Not decomposable (decomposable = 0; arrays unspecified) also when
//   a (step, p, q) occurs twice -- the row map could hold only one of them.
AF_EXPORT int af_fused_plan_antennas(const int64_t *time_index_host, const int32_t *antenna1_host,
                                     const int32_t *antenna2_host, const double *uvw_host, int64_t nrow, int64_t nant,
                                     double tol, int64_t nsteps, double *ant_uvw_host, int32_t *rowmap_host,
                                     double *max_residual, int *decomposable)
{
    AF_REQUIRE(decomposable != nullptr && max_residual != nullptr, "af_fused_plan_antennas: result pointer is NULL");
    *decomposable = 0;
    *max_residual = 0.0;
    if (nrow == 0) { *decomposable = 1; return AF_OK; }
    AF_REQUIRE(time_index_host && antenna1_host && antenna2_host && uvw_host && ant_uvw_host && rowmap_host,
               "af_fused_plan_antennas: NULL array");
    AF_REQUIRE(nant >= 1 && nant <= 512 && nrow < (1LL << 31) && nsteps >= 1, "af_fused_plan_antennas: bad extents");
    int64_t tmin = time_index_host[0], tmax = tmin;
    for (int64_t r = 1; r < nrow; ++r) {
        tmin = time_index_host[r] < tmin ? time_index_host[r] : tmin;
        tmax = time_index_host[r] > tmax ? time_index_host[r] : tmax;
    }
    AF_REQUIRE(tmax - tmin + 1 == nsteps, "af_fused_plan_antennas: nsteps %lld, time_index spans %lld", (long long)nsteps,
               (long long)(tmax - tmin + 1));
    const int64_t nap = 8 * ((nant + 7) / 8);
    for (int64_t i = 0; i < nsteps * nap * nap; ++i) rowmap_host[i] = -1;
    for (int64_t i = 0; i < nsteps * nant * 3; ++i) ant_uvw_host[i] = 0.0;
    // rows by step (counting sort) and the row map
    std::vector<int64_t> first((size_t)nsteps + 1, 0);
    for (int64_t r = 0; r < nrow; ++r) ++first[(size_t)(time_index_host[r] - tmin) + 1];
    for (int64_t s = 0; s < nsteps; ++s) first[(size_t)s + 1] += first[(size_t)s];
    std::vector<int32_t> order((size_t)nrow);
    {
        std::vector<int64_t> at(first.begin(), first.end() - 1);
        for (int64_t r = 0; r < nrow; ++r) order[(size_t)at[(size_t)(time_index_host[r] - tmin)]++] = (int32_t)r;
    }
    for (int64_t r = 0; r < nrow; ++r) {
        const int32_t p = antenna1_host[r], q = antenna2_host[r];
        AF_REQUIRE(p >= 0 && p < nant && q >= 0 && q < nant, "af_fused_plan_antennas: antenna index out of range");
        int32_t &slot = rowmap_host[((time_index_host[r] - tmin) * nap + p) * nap + q];
        if (slot != -1) return AF_OK;             // the same baseline twice in one step
        slot = (int32_t)r;
    }
    double worst = 0.0;
    std::vector<int> comp((size_t)nant), queue;
    std::vector<std::vector<int32_t>> adj((size_t)nant);
    for (int64_t s = 0; s < nsteps; ++s) {
        double *x = ant_uvw_host + s * nant * 3;
        for (auto &a : adj) a.clear();
        for (int64_t k = first[(size_t)s]; k < first[(size_t)s + 1]; ++k) {
            const int32_t r = order[(size_t)k];
            adj[(size_t)antenna1_host[r]].push_back(r);
            if (antenna2_host[r] != antenna1_host[r]) adj[(size_t)antenna2_host[r]].push_back(r);
        }
        std::fill(comp.begin(), comp.end(), -1);
        int ncomp = 0;
        for (int root = 0; root < (int)nant; ++root) {
            if (comp[(size_t)root] >= 0 || adj[(size_t)root].empty()) continue;
            comp[(size_t)root] = ncomp;
            queue.assign(1, root);
            for (size_t h = 0; h < queue.size(); ++h) {
                const int p = queue[h];
                for (int32_t r : adj[(size_t)p]) {
                    const int a1 = antenna1_host[r], a2 = antenna2_host[r];
                    const int o = a1 == p ? a2 : a1;
                    if (comp[(size_t)o] >= 0) continue;
                    comp[(size_t)o] = ncomp;
                    for (int c = 0; c < 3; ++c)      // uvw_r = x_a1 - x_a2
                        x[3 * o + c] = a1 == p ? x[3 * p + c] - uvw_host[3 * (int64_t)r + c] : x[3 * p + c] + uvw_host[3 * (int64_t)r + c];
                    queue.push_back(o);
                }
            }
            ++ncomp;
        }
        for (int sweep = 0; sweep < 2; ++sweep)
            for (int p = 0; p < (int)nant; ++p) {
                double acc[3] = {0.0, 0.0, 0.0};
                int n = 0;
                for (int32_t r : adj[(size_t)p]) {
                    const int a1 = antenna1_host[r], a2 = antenna2_host[r];
                    if (a1 == a2) continue;
                    const int o = a1 == p ? a2 : a1;
                    for (int c = 0; c < 3; ++c)
                        acc[c] += a1 == p ? x[3 * o + c] + uvw_host[3 * (int64_t)r + c] : x[3 * o + c] - uvw_host[3 * (int64_t)r + c];
                    ++n;
                }
                if (n)
                    for (int c = 0; c < 3; ++c) x[3 * p + c] = acc[c] / n;
            }
        for (int k = 0; k < ncomp; ++k) {          // remove every component's mean
            double mean[3] = {0.0, 0.0, 0.0};
            int n = 0;
            for (int p = 0; p < (int)nant; ++p)
                if (comp[(size_t)p] == k) { for (int c = 0; c < 3; ++c) mean[c] += x[3 * p + c]; ++n; }
            for (int p = 0; p < (int)nant; ++p)
                if (comp[(size_t)p] == k) for (int c = 0; c < 3; ++c) x[3 * p + c] -= mean[c] / n;
        }
        for (int64_t k = first[(size_t)s]; k < first[(size_t)s + 1]; ++k) {
            const int32_t r = order[(size_t)k];
            const int a1 = antenna1_host[r], a2 = antenna2_host[r];
            for (int c = 0; c < 3; ++c) {
                const double d = fabs(x[3 * a1 + c] - x[3 * a2 + c] - uvw_host[3 * (int64_t)r + c]);
                if (!(d <= worst)) worst = d;       // NaN sticks
            }
        }
    }
    *max_residual = worst;
    *decomposable = worst <= tol ? 1 : 0;
    return AF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Plan guard.  A plan (af_fused_plan_groups / _rows / _antennas) is made from one call's (time_index, antenna1,
// antenna2 [, uvw]); the predict entries then read the PLAN's arrays, not the call's.  af_fused_plan_check proves on
// the device, in O(row) and without a host round trip, that a plan belongs to the arrays of THIS call:
//   * row r of the call has the plan's step (up to the common offset time_index[0] - plan_step[0]) and antennas;
//   * with ant_uvw: |x_a1 - x_a2 - uvw_r|_inf <= tol  (x = the plan's antenna coordinates of the row's step).
// A mismatch sets AF_STATUS_PLAN_INDEX / AF_STATUS_PLAN_UVW in *status and fills `out` with NaN (the predict has
// already run on the same stream: the call returns nothing that looks like a result).
namespace {

template <typename IT>
__global__ void plan_check_kernel(const IT *__restrict__ time_index, const IT *__restrict__ antenna1,
                                  const IT *__restrict__ antenna2, const double *__restrict__ uvw, int64_t nrow,
                                  const int32_t *__restrict__ plan_step, const int32_t *__restrict__ plan_a1,
                                  const int32_t *__restrict__ plan_a2, const double *__restrict__ ant_uvw, int64_t nant,
                                  double tol, int *__restrict__ status)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    int flags = 0;
    const int64_t shift = (int64_t)time_index[0] - plan_step[0];
    const int64_t step = plan_step[r];
    const int a1 = plan_a1[r], a2 = plan_a2[r];
    if ((int64_t)time_index[r] - shift != step || (int64_t)antenna1[r] != a1 || (int64_t)antenna2[r] != a2)
        flags |= AF_STATUS_PLAN_INDEX;
    if (ant_uvw != nullptr && uvw != nullptr) {
        const double *x1 = ant_uvw + (step * nant + a1) * 3, *x2 = ant_uvw + (step * nant + a2) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (!(fabs(x1[c] - x2[c] - uvw[3 * r + c]) <= tol)) flags |= AF_STATUS_PLAN_UVW;     // NaN fails
    }
    if (flags) atomicOr(status, flags);
}

__global__ void plan_poison_kernel(const int *__restrict__ status, double *__restrict__ out, int64_t n)
{
    if (*status == 0) return;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = nan;
}

}  // namespace

AF_EXPORT int af_fused_plan_check(const void *time_index, const void *antenna1, const void *antenna2, int index_bytes,
                                  const double *uvw, int64_t nrow, const int32_t *plan_step, const int32_t *plan_antenna1,
                                  const int32_t *plan_antenna2, const double *ant_uvw, int64_t nant, double tol,
                                  double *out, int64_t out_doubles, int32_t *status, void *stream)
{
    AF_REQUIRE(index_bytes == 4 || index_bytes == 8, "af_fused_plan_check: index_bytes must be 4 or 8");
    AF_REQUIRE(status != nullptr, "af_fused_plan_check: status is NULL");
    AF_REQUIRE(nrow >= 0 && nant >= 0 && out_doubles >= 0, "af_fused_plan_check: negative extent");
    hipStream_t st_ = af_stream(stream);
    AF_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st_));
    if (nrow == 0) return AF_OK;
    AF_REQUIRE(time_index && antenna1 && antenna2 && plan_step && plan_antenna1 && plan_antenna2,
               "af_fused_plan_check: NULL array");
    AF_REQUIRE((ant_uvw == nullptr) || uvw != nullptr, "af_fused_plan_check: ant_uvw without uvw");
    const unsigned blocks = (unsigned)af_cdiv(nrow, 256);
    if (index_bytes == 4)
        hipLaunchKernelGGL(plan_check_kernel<int32_t>, dim3(blocks), dim3(256), 0, st_, (const int32_t *)time_index,
                           (const int32_t *)antenna1, (const int32_t *)antenna2, uvw, nrow, plan_step, plan_antenna1,
                           plan_antenna2, ant_uvw, nant, tol, status);
    else
        hipLaunchKernelGGL(plan_check_kernel<int64_t>, dim3(blocks), dim3(256), 0, st_, (const int64_t *)time_index,
                           (const int64_t *)antenna1, (const int64_t *)antenna2, uvw, nrow, plan_step, plan_antenna1,
                           plan_antenna2, ant_uvw, nant, tol, status);
    AF_LAUNCH_CHECK();
    if (out != nullptr && out_doubles > 0) {
        hipLaunchKernelGGL(plan_poison_kernel, dim3(2048), dim3(256), 0, st_, status, out, out_doubles);
        AF_LAUNCH_CHECK();
    }
    return AF_OK;
}

// How the 8-antenna blocks of M are cut into super-blocks (host): runs of 8 blocks, the last one shorter.
struct GemmTiling {
    int nsb;
    int size[8], blk0[8];
};
constexpr int RECT_COLS = 4;      // column blocks of a RECT super-tile (rows: 8)
static void gemm_tiling(int nb, GemmTiling &tl)
{
    tl.nsb = (nb + 7) / 8;
    for (int i = 0; i < tl.nsb; ++i) {
        tl.size[i] = nb - 8 * i < 8 ? nb - 8 * i : 8;
        tl.blk0[i] = 8 * i;
    }
}

// Baseline slots (8 x 8-antenna tiles x 64) the GEMM form evaluates per (timestep, channel) for `nant` antennas: what its
// cost is proportional to (the caller's fill-factor rule divides the rows per step by it).  0 beyond 512 antennas.
AF_EXPORT int64_t af_fused_gemm_slots(int64_t nant)
{
    if (nant < 1 || nant > 512) return 0;
    GemmTiling tl;
    gemm_tiling((int)((nant + 7) / 8), tl);
    int64_t tiles = 0;
    for (int i = 0; i < tl.nsb; ++i) {
        tiles += tl.size[i] * (tl.size[i] + 1) / 2;
        for (int j = i + 1; j < tl.nsb; ++j) tiles += tl.size[i] * tl.size[j];
    }
    return tiles * 64;
}

// The antenna-decomposed form of af_fused_predict_c128: same arguments, with the plan of af_fused_plan_antennas
// (DEVICE copies: ant_uvw (nsteps, nant, 3), rowmap (nsteps, nap, nap)) in place of uvw, antenna1 / antenna2 and the
// items; nsteps <= ntime; every row of `out` that the row map names is written, nothing else is touched.  Same workspace
// as af_fused_predict_c128 (af_fused_predict_workspace_bytes).  nant <= 512 (64 blocks of 8 antennas in <= 8 super-blocks);
// no gauss_shape.
AF_EXPORT int af_fused_predict_antennas_c128(const double *ant_uvw, const int32_t *rowmap, int64_t nsteps, int64_t nrow,
                                             const double *lm, const double *frequency, const double *brightness,
                                             int64_t nsrc, int64_t nchan, const double *beam, int64_t beam_lw,
                                             int64_t beam_mh, int64_t beam_nud, const double *beam_lm_extents,
                                             const double *beam_freq_map, const double *parallactic_angles, int64_t ntime,
                                             int64_t nant, const double *point_errors, const double *antenna_scaling,
                                             const double *feed_rotation, int convention, double *out, void *workspace,
                                             size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    AF_REQUIRE(beam_lw >= 2 && beam_mh >= 2 && beam_nud >= 2, "beam_lw, beam_mh and beam_nud must be >= 2");
    AF_REQUIRE(nsteps >= 0 && nrow >= 0 && nsrc >= 0 && nchan >= 0 && ntime >= 0 && nant >= 0,
               "af_fused_predict_antennas_c128: negative extent");
    AF_REQUIRE(nant <= 512, "af_fused_predict_antennas_c128: more than 512 antennas (use af_fused_predict_c128)");
    AF_REQUIRE(nsteps <= ntime, "af_fused_predict_antennas_c128: %lld steps but %lld timesteps of per-antenna terms",
               (long long)nsteps, (long long)ntime);
    AF_REQUIRE(nsrc < (1LL << 31) && nchan <= 65535 && nsteps < (1LL << 31), "af_fused_predict_antennas_c128: too large");
    hipStream_t st_ = af_stream(stream);
    if (nrow == 0 || nchan == 0 || nsteps == 0) return AF_OK;
    AF_REQUIRE(out != nullptr, "af_fused_predict_antennas_c128: out is NULL");
    if (nsrc == 0) {
        AF_HIP(hipMemsetAsync(out, 0, sizeof(double) * 2 * 4 * (size_t)(nrow * nchan), st_));
        return AF_OK;
    }
    AF_REQUIRE(ant_uvw && rowmap && lm && frequency && brightness && beam && beam_lm_extents && beam_freq_map &&
                   parallactic_angles && point_errors && antenna_scaling,
               "af_fused_predict_antennas_c128: NULL array");
    const FusedWs W = fused_ws(nsrc, nchan, beam_lw, beam_mh, beam_nud);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= W.total,
               "af_fused_predict_antennas_c128: workspace too small (%zu < %zu)", workspace_bytes, W.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_fused_predict_antennas_c128: workspace must be 256-byte aligned");
    char *ws = static_cast<char *>(workspace);
    double *lmn = reinterpret_cast<double *>(ws + W.lmn), *f4 = reinterpret_cast<double *>(ws + W.f4);
    double *freq_data = reinterpret_cast<double *>(ws + W.freq_data), *planes_buf = reinterpret_cast<double *>(ws + W.planes);
    hipLaunchKernelGGL(fused_prep_src, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st_, lm, nsrc, lmn);
    AF_LAUNCH_CHECK();
    hipLaunchKernelGGL(fused_prep_freq, dim3((unsigned)af_cdiv(nchan, 64)), dim3(64), 0, st_, frequency, nchan,
                       convention, f4);
    AF_LAUNCH_CHECK();
    int rc = af_freq_grid_interp_f64(frequency, nchan, beam_freq_map, beam_nud, freq_data, stream);
    if (rc != AF_OK) return rc;
    const int64_t ncell = beam_lw * beam_mh;
    AF_REQUIRE(ncell < (1LL << 25), "af_fused_predict_antennas_c128: beam cube too large (fewer than 2^25 cells per plane)");
    // profiling builds only (make HOOKS=1 -> lib/prof/libafhip.so, -DAFHIP_STAGE_HOOKS): run one stage / change the
    // sampling waves' priority; the shipped library does not read the environment here
    static const int only_stage = AF_STAGE_ENV("AFHIP_FUSED_STAGE", 0);
    static const int sample_prio = AF_STAGE_ENV("AFHIP_GEMM_PRIO", 1);
    const bool feed = feed_rotation != nullptr;
    const int nb = (int)((nant + 7) / 8), nap = 8 * nb;
    // the super-tiles of M (see Geo): one DIAG for nant <= 64; beyond that super-blocks of 4 .. 6 blocks
    GemmTiling tl;
    gemm_tiling(nb, tl);
    // one main-kernel launch per (channel group, super-tile shape)
    struct Shape {
        const void *kernel;
        size_t lds;
        SuperTileList list;
        int count;
    };
    std::vector<Shape> shapes;
    auto add = [&](const void *kernel, size_t lds, const SuperTile &e) {
        for (auto &s : shapes)
            if (s.kernel == kernel && s.count < 16) { s.list.e[s.count++] = e; return; }
        Shape s;
        memset(&s, 0, sizeof(s));
        s.kernel = kernel; s.lds = lds; s.list.e[0] = e; s.count = 1;
        shapes.push_back(s);
    };
    // sources per batch: about one super-round of the 256 sampling lanes, the panel buffers within ~130 KB
    // which of the four sampling waves run order B (bit w = wave 8 + w; see the kernel): waves 10 and 11 by default.
    // AFHIP_GEMM_ORDER_B = another mask (A/B hook between correct schedules; 0 = round 4's order for all)
    // (profiling build only, like the other stage hooks: the shipped library does not read the environment; the delay is
    // clamped -- every matrix wave sleeps it on every batch)
    static const int order_b_env = AF_STAGE_ENV("AFHIP_GEMM_ORDER_B", 0) & 15;
    static const int burst_delay_env = AF_STAGE_ENV("AFHIP_GEMM_BURST_DELAY", 0);   // 64-cycle units
    int order_b_mask = order_b_env;
    int burst_delay = burst_delay_env < 0 ? 0 : (burst_delay_env > 4096 ? 4096 : burst_delay_env);
#define AF_GEMM_K(RECTC, NBRC, NBCC, STC)                                                                              \
    (feed ? reinterpret_cast<const void *>(fused_gemm3_kernel<true, RECTC, NBRC, NBCC, STC>)                            \
          : reinterpret_cast<const void *>(fused_gemm3_kernel<false, RECTC, NBRC, NBCC, STC>)),                         \
        Geo<RECTC, NBRC, NBCC, STC>::lds_bytes()
#define AF_GEMM_K4(NBRC, NBCC, STC)                                                                                    \
    (feed ? reinterpret_cast<const void *>(fused_gemm3_kernel<true, true, NBRC, NBCC, STC, true>)                       \
          : reinterpret_cast<const void *>(fused_gemm3_kernel<false, true, NBRC, NBCC, STC, true>)),                    \
        Geo<true, NBRC, NBCC, STC, true>::lds_bytes()
    // pairs of super-blocks: a column super-block of more than 4 blocks takes ONE 8 x 8 super-tile in the four-product form
    // (AFHIP_GEMM_RECT8=0 in the profiling build: round 5's two 8 x 4 super-tiles, for A/B runs)
    static const int rect8 = AF_STAGE_ENV("AFHIP_GEMM_RECT8", 1);
    for (int i = 0; i < tl.nsb; ++i) {
        SuperTile e = {8 * tl.blk0[i], 8 * tl.blk0[i], tl.size[i], 0};
        switch (tl.size[i]) {
        case 1: add(AF_GEMM_K(false, 1, 1, 16), e); break;
        case 2: add(AF_GEMM_K(false, 2, 2, 8), e); break;
        case 3: add(AF_GEMM_K(false, 3, 3, 8), e); break;
        case 4: add(AF_GEMM_K(false, 4, 4, 6), e); break;
        case 5: add(AF_GEMM_K(false, 5, 5, 4), e); break;
        case 6: add(AF_GEMM_K(false, 6, 6, 4), e); break;
        case 7: add(AF_GEMM_K(false, 7, 7, 4), e); break;
        default: add(AF_GEMM_K(false, 8, 8, 4), e); break;
        }
        // the pairs with the later super-blocks: rows = this (full, 8-block) super-block, columns = the later one in
        // chunks of <= 4 blocks (8 x 4 tiles: 4 per matrix wave, wave w = block row w)
        for (int j = i + 1; j < tl.nsb; ++j) {
            if (rect8 && tl.size[j] > RECT_COLS) {
                SuperTile r = {8 * tl.blk0[i], 8 * tl.blk0[j], tl.size[j], 0};
                add(AF_GEMM_K4(8, 8, 4), r);
                continue;
            }
            for (int c0 = 0; c0 < tl.size[j]; c0 += RECT_COLS) {
                SuperTile r = {8 * tl.blk0[i], 8 * (tl.blk0[j] + c0), tl.size[j] - c0 < RECT_COLS ? tl.size[j] - c0 : RECT_COLS, 0};
                add(AF_GEMM_K(true, 8, RECT_COLS, 4), r);
            }
        }
    }
#undef AF_GEMM_K
#undef AF_GEMM_K4
    for (auto &s : shapes) {
        AF_REQUIRE(s.lds <= 160 * 1024, "af_fused_predict_antennas_c128: %zu bytes of LDS needed", s.lds);
        AF_HIP(hipFuncSetAttribute(s.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lds));
    }
    int nsrc_i = (int)nsrc, nant_i = (int)nant, nap_i = nap, stage_i = only_stage, prio_i = sample_prio;
    const double2 *b2 = reinterpret_cast<const double2 *>(brightness), *fr2 = reinterpret_cast<const double2 *>(feed_rotation);
    double2 *out2 = reinterpret_cast<double2 *>(out);
    for (int64_t f0 = 0; f0 < nchan; f0 += PLANE_GROUP) {
        const int64_t nf = nchan - f0 < PLANE_GROUP ? nchan - f0 : PLANE_GROUP;
        int64_t blocks = af_cdiv(ncell * 4, 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(beam_plane_kernel, dim3((unsigned)blocks, (unsigned)nf), dim3(256), 0, st_,
                           reinterpret_cast<const double2 *>(beam), ncell, beam_nud, freq_data, f0, planes_buf);
        AF_LAUNCH_CHECK();
        if (f0 == 0) af_prof_begin(st_);
        for (auto &s : shapes) {
            void *args[] = {&ant_uvw, &rowmap, &lmn, &f4, &b2, &planes_buf, &beam_lw, &beam_mh, &beam_nud, &beam_lm_extents,
                            &freq_data, &parallactic_angles, &point_errors, &antenna_scaling, &fr2, &nsrc_i, &nchan, &ntime,
                            &nant_i, &nap_i, &out2, &stage_i, &f0, &prio_i, &order_b_mask, &burst_delay, &s.list};
            AF_HIP(hipLaunchKernel(s.kernel, dim3((unsigned)nsteps, (unsigned)nf, (unsigned)s.count), dim3(G3_THREADS), args,
                                   s.lds, st_));
        }
        if (f0 == 0) af_prof_end(st_);
        AF_LAUNCH_CHECK();
    }
    return AF_OK;
}
