// Fused RIME predict with per-antenna beam-cube DDEs for antenna-decomposable uvw, SINGLE PRECISION: every input float32 /
// complex64, complex64 out -- the precision in which the reference runs its chain for such callers
// (africanus/util/type_inference.py:24-26: the promoted input type; africanus/rime/predict.py:542-544;
// africanus/rime/phase.py:28-61 and africanus/rime/fast_beam_cubes.py:57-240 with float32 arguments).
//
// Same formulation as af_fused_gemm.hip -- V(t, nu) = G H^H per (timestep, channel), G_a = k_a E_a X_s, H_a = k_a E_a --
// re-designed for what single precision changes on this machine:
//   * v_mfma_f32_16x16x4_f32 runs at twice the fp64 rate on its own pipe, and the sampling waves' arithmetic (float32 VALU)
//     no longer queues behind the matrix waves' instructions: the two halves of a batch overlap instead of adding up;
//   * the per-channel beam planes are 64-byte records of float (re, im, |.|, 0) x 4 correlations: one 12-byte load per
//     correlation and corner instead of a 16- and an 8-byte one, and the (l, m) / (l, m + 1) corners share a 128-byte line
//     half of the time;
//   * operand panels in LDS are float: 4 H planes + 4 G planes per source (re / im x the two Jones columns), 2.7 x smaller than
//     the fp64 kernel's six + six, so a RECT super-tile's batch is a whole number of super-rounds (no flat term stream, two
//     panel buffers);
//   * the complex product in its direct four-product form (Re += Gr Hr + Gi Hi, Im += Gi Hr, Im' += Gr Hi; Im - Im' in the
//     epilogue): the matrix pipe is not what bounds this kernel, the sampler's LDS stores are -- the 3M form would add two
//     planes to every term.
// What stays double: the antenna phasor's argument and the phasor itself (|phase| reaches 1e4 rad: float32 would lose the
// result, as the reference's float32 phase_delay does), the voxel coordinates (four FMAs per term) and the per-slot
// coefficients they come from.  The entry is therefore CLOSER to the float64 chain on the same float32 inputs than the
// reference's own float32 chain (tests/test_gpu_fused_gemm_c64.py: golden G17, recorded from the reference).
//
// Workgroup: 16 waves at <= 128 registers -- 8 matrix waves (<= 5 tiles of 8 x 8 antennas each, compile-time tile
// coordinates) and EIGHT sampling waves (quad = the four correlations of one (source, antenna) term, four sampling rounds per
// super-round of 512 terms as in af_fused_predict.hip): a batch lasts as long as a sampling wave's serial chain (coordinates
// -> gathers -> arithmetic -> panel stores), and single precision leaves the registers for twice the fp64 kernel's four
// (first version, 12 waves: 68.9 ms for 1e6 rows x 64 channels x 1000 sources against the fp64 kernel's 87.4).
// Super-tiles (DIAG / RECT 8 x 4 / RECT 8 x 8) and the row map as in af_fused_gemm.hip.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "af_fused_device_f32.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int S_THREADS = 1024, S_MATRIX = 512, S_SAMPLERS = 512, S_PLANES = 4;

// doubles of ... floats of padding per operand plane such that (plane stride) mod 32 == 16: the two Jones columns a
// matrix wave's lanes 0-15 / 16-31 read with one ds_read_b32 then sit on disjoint halves of the 32 banks
constexpr int plane_stride(int nant) { return 2 * nant + ((2 * nant) % 32 == 0 ? 16 : 32); }

// Three forms of the complex product, chosen per super-tile shape:
//   FORM_4P  (RECT 8 x 4) four products on three accumulators, Im = sum Gi Hr - sum Gr Hi in the epilogue; planes re, im.
//   FORM_3M  (DIAG)       P1 = Gr Hr, P2 = Gi Hi, P3 = (Gr + Gi)(Hr - Hi): three matrix instructions per tile and source pair
//                         for two more planes per term (re -+ im) -- 54.6 -> 47.2 ms at 64 antennas, where the matrix
//                         instructions are half of what a SIMD issues; a RECT 8 x 4 batch of whole super-rounds would not fit
//                         the LDS with six + six planes.
//   FORM_ROW (RECT 8 x 8) the 3M form with matrix wave W = block row W: eight tiles x three accumulators are 96 float
//                         registers (the fp64 kernel needs its two-accumulator FOURM form for this shape; here the 3M form
//                         fits: 116 registers in all) -- one super-tile per pair of super-blocks, 128 sampled terms for 64
//                         tiles instead of two 8 x 4 super-tiles' 192.
constexpr int FORM_4P = 0, FORM_3M = 1, FORM_ROW = 2;
template <bool RECT, int NBR, int NBC, int ST, int FORM = FORM_4P>
struct GeoS {
    static constexpr int NAR = 8 * NBR, NAC = 8 * NBC;
    static constexpr int CSG = plane_stride(NAR), CSH = plane_stride(NAC);     // floats per operand plane
    static constexpr int HP = FORM == FORM_4P ? S_PLANES : 6, GPL = HP;               // H / G planes of a source
    static constexpr int SRC = HP * CSH + GPL * CSG;                           // one source: H planes, then G planes
    static constexpr int BUF = ST * SRC;                                       // one batch
    static constexpr int TPS = RECT ? NAR + NAC : NAR;                         // sampled terms (antenna slots) per source
    static constexpr int BT = ST * TPS;                                        // terms per batch
    static constexpr int SRB = (BT + S_SAMPLERS - 1) / S_SAMPLERS;             // super-rounds per batch (the last one padded)
    static constexpr int NTILE = RECT ? NBR * NBC : NBR * (NBR + 1) / 2;
    static constexpr size_t lds_bytes()
    {
        return (size_t)2 * BUF * sizeof(float) + (size_t)TPS * (6 + 4) * sizeof(double) + (size_t)TPS * 4 * sizeof(float2) +
               PH_TABLE * sizeof(double2);
    }
};

struct SuperTileS {
    int row_ant0, col_ant0;   // first antenna of the row / column super-block
    int nc_act;               // RECT: column blocks actually present (1 .. NBC)
    int pad_;
};
struct SuperTileListS {
    SuperTileS e[16];
};

// tiles of the upper block triangle in row-pair order (consecutive tiles of a wave share their block row)
constexpr int pair_tile_s(int nb, int idx, bool col)
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
template <bool RECT, int NBR, int NBC> constexpr int tile_row_s(int idx) { return RECT ? idx / NBC : pair_tile_s(NBR, idx, false); }
template <bool RECT, int NBR, int NBC> constexpr int tile_col_s(int idx) { return RECT ? idx % NBC : pair_tile_s(NBR, idx, true); }
constexpr int chunk_lo_s(int ntile, int c) { return c * ntile / 8; }

template <typename F, int... Js>
__device__ __forceinline__ void for_each_const_s(F &&fn, std::integer_sequence<int, Js...>)
{
    (fn(std::integral_constant<int, Js>{}), ...);
}

// Operand element of antenna slot `ant` inside its panel of 4 q4 antennas, and back.  In sampling round QL the sixteen
// quads of a wave store the terms of slots QL, 4 + QL, 8 + QL, ...: with the antennas in their own order those stores
// would land 4-way on the same banks (element stride 8 floats; profiles/r06_fused_dde_ant_c64_pmc_summary.json of the first
// version: 42 % of the LDS cycles were conflicts).  Slot 4 i + QL -> element QL q4 + i: the quads of one store are
// consecutive elements.  Tiles are blocks of 8 consecutive ELEMENTS; the epilogue maps them back.
__device__ __forceinline__ int slot_antenna_s(int slot, int q4) { return (slot % q4) * 4 + slot / q4; }
__device__ __forceinline__ int antenna_slot_s(int ant, int q4) { return (ant & 3) * q4 + (ant >> 2); }


// ---- matrix waves ------------------------------------------------------------------------------------------------------
// D_LAYOUT_STD: accumulator register r of lane l holds D[4 (l >> 4) + r][l & 15] (v_mfma_f32_16x16x4_f32;
// tools/probe/probe_mfma_f32_16x16x4.hip prints it).
template <bool RECT, int NBR, int NBC, int ST, int FORM, int W>
__device__ __forceinline__ void matrix_wave_s(const float *__restrict__ lds, int nbatch, int lane,
                                              const int32_t *__restrict__ rm, int nap, const SuperTileS tile, int64_t nchan,
                                              int64_t f, float2 *__restrict__ out)
{
    using G = GeoS<RECT, NBR, NBC, ST, FORM>;
    constexpr int CSG = G::CSG, CSH = G::CSH, SRC = G::SRC, BUF = G::BUF;
    constexpr int T0 = chunk_lo_s(G::NTILE, W), CNT = chunk_lo_s(G::NTILE, W + 1) - T0;
    constexpr int CN = CNT > 0 ? CNT : 1;
    const int kq = lane >> 4, c16 = lane & 15;
    // K index kq = 2 (source of the pair) + (column of the Jones row): this lane's element of every operand plane
    const int offB = (kq >> 1) * SRC + (kq & 1) * CSH + c16;
    const int offA = (kq >> 1) * SRC + G::HP * CSH + (kq & 1) * CSG + c16;
    v4f cr[CN], ci1[CN], ci2[CN];
#pragma unroll
    for (int j = 0; j < CN; ++j) cr[j] = ci1[j] = ci2[j] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    int buf = 0;
    for (int b = 0; b < nbatch; ++b) {
        __syncthreads();
        const float *P = lds + buf * BUF;
        buf ^= 1;
        if (CNT == 0) continue;
#ifdef AF_C64_STAGE_SAMPLE        // (diagnostic build: the sampling waves alone; the output is meaningless)
        continue;
#endif
#pragma unroll
        for (int s2 = 0; s2 < ST / 2; ++s2) {
            const float *S = P + 2 * s2 * SRC;
            for_each_const_s([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int pb = tile_row_s<RECT, NBR, NBC>(T0 + j), qb = tile_col_s<RECT, NBR, NBC>(T0 + j);
                if (RECT && qb >= tile.nc_act) return;           // block-uniform: the column super-block is short
                const float *A = S + offA + pb * 16, *B = S + offB + qb * 16;
if constexpr (FORM == FORM_3M) {
                    cr[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0], B[0], cr[j], 0, 0, 0);                       // P1 = Gr Hr
                    ci1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[2 * CSG], B[2 * CSH], ci1[j], 0, 0, 0);         // P2 = Gi Hi
                    ci2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[4 * CSG], B[4 * CSH], ci2[j], 0, 0, 0);         // P3 = (Gr + Gi)(Hr - Hi)
                } else {
                    const float gr = A[0], gi = A[2 * CSG], hr = B[0], hi = B[2 * CSH];
                    cr[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr, hr, cr[j], 0, 0, 0);
                    ci1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gi, hr, ci1[j], 0, 0, 0);
                    ci2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr, hi, ci2[j], 0, 0, 0);
                    cr[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gi, hi, cr[j], 0, 0, 0);
                }
            }, std::make_integer_sequence<int, CNT>{});
        }
    }
    for_each_const_s([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int pb = tile_row_s<RECT, NBR, NBC>(T0 + j), qb = tile_col_s<RECT, NBR, NBC>(T0 + j);
        if (RECT && qb >= tile.nc_act) return;
        const int jj = c16 & 1;
        const int q = tile.col_ant0 + slot_antenna_s(qb * 8 + (c16 >> 1), RECT ? 2 * tile.nc_act : G::NAC / 4);
        int r1[4], r2[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = 4 * kq + reg;                        // D row of accumulator register `reg`
            const int p = tile.row_ant0 + slot_antenna_s(pb * 8 + (i >> 1), G::NAR / 4);
            r1[reg] = rm[p * nap + q];
            r2[reg] = (RECT || pb != qb) ? rm[q * nap + p] : -1;   // the same antennas the other way round: V_qp = V_pq^H
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ii = (4 * kq + reg) & 1;
const float re = FORM == FORM_3M ? cr[j][reg] + ci1[j][reg] : cr[j][reg];
            const float im = FORM == FORM_3M ? (ci2[j][reg] - cr[j][reg]) + ci1[j][reg] : ci1[j][reg] - ci2[j][reg];
            if (r1[reg] >= 0) out[((int64_t)r1[reg] * nchan + f) * 4 + ii * 2 + jj] = make_float2(re, im);
            if (r2[reg] >= 0) out[((int64_t)r2[reg] * nchan + f) * 4 + jj * 2 + ii] = make_float2(re, -im);
        }
    }, std::make_integer_sequence<int, CNT>{});
}

// ROW form (RECT 8 x NBC): wave W = block row W, 3M on three accumulators per tile
template <int NBC, int ST, int W>
__device__ __forceinline__ void matrix_wave_row(const float *__restrict__ lds, int nbatch, int lane, const int32_t *__restrict__ rm,
                                                int nap, const SuperTileS tile, int64_t nchan, int64_t f, float2 *__restrict__ out)
{
    using G = GeoS<true, 8, NBC, ST, FORM_ROW>;
    constexpr int CSG = G::CSG, CSH = G::CSH, SRC = G::SRC, BUF = G::BUF;
    const int kq = lane >> 4, c16 = lane & 15;
    const int offB = (kq >> 1) * SRC + (kq & 1) * CSH + c16;
    const int offA = (kq >> 1) * SRC + G::HP * CSH + (kq & 1) * CSG + c16 + W * 16;
    v4f p1[NBC], p2[NBC], p3[NBC];                       // 3M: P1 = Gr Hr, P2 = Gi Hi, P3 = (Gr + Gi)(Hr - Hi)
#pragma unroll
    for (int j = 0; j < NBC; ++j) p1[j] = p2[j] = p3[j] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    int buf = 0;
    for (int b = 0; b < nbatch; ++b) {
        __syncthreads();
        const float *P = lds + buf * BUF;
        buf ^= 1;
#ifdef AF_C64_STAGE_SAMPLE
        continue;
#endif
#pragma unroll
        for (int s2 = 0; s2 < ST / 2; ++s2) {
            const float *A = P + 2 * s2 * SRC + offA, *Bp = P + 2 * s2 * SRC + offB;
            const float gr = A[0], gi = A[2 * CSG], gs = A[4 * CSG];      // the block row's operands: once per source pair
            for_each_const_s([&](auto jc) {
                constexpr int qb = decltype(jc)::value;
                if (qb >= tile.nc_act) return;                   // block-uniform: the column super-block is short
                const float *B = Bp + qb * 16;
                p1[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr, B[0], p1[qb], 0, 0, 0);
                p2[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gi, B[2 * CSH], p2[qb], 0, 0, 0);
                p3[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gs, B[4 * CSH], p3[qb], 0, 0, 0);
            }, std::make_integer_sequence<int, NBC>{});
        }
    }
    for_each_const_s([&](auto jc) {
        constexpr int qb = decltype(jc)::value;
        if (qb >= tile.nc_act) return;
        const int jj = c16 & 1;
        const int q = tile.col_ant0 + slot_antenna_s(qb * 8 + (c16 >> 1), 2 * tile.nc_act);
        int r1[4], r2[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int p = tile.row_ant0 + slot_antenna_s(W * 8 + ((4 * kq + reg) >> 1), G::NAR / 4);
            r1[reg] = rm[p * nap + q];
            r2[reg] = rm[q * nap + p];
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ii = (4 * kq + reg) & 1;
            const float re = p1[qb][reg] + p2[qb][reg], im = (p3[qb][reg] - p1[qb][reg]) + p2[qb][reg];
            if (r1[reg] >= 0) out[((int64_t)r1[reg] * nchan + f) * 4 + ii * 2 + jj] = make_float2(re, im);
            if (r2[reg] >= 0) out[((int64_t)r2[reg] * nchan + f) * 4 + jj * 2 + ii] = make_float2(re, -im);
        }
    }, std::make_integer_sequence<int, NBC>{});
}

// grid: (nsteps, channels of the plane group, super-tiles of this shape); block 1024.
template <bool FEED, bool RECT, int NBR, int NBC, int ST, int FORM>
__global__ __launch_bounds__(S_THREADS) void fused_gemm_c64_kernel(
    const double *__restrict__ ant_uvw, const int32_t *__restrict__ rowmap, const double *__restrict__ lmn,
    const double *__restrict__ f4, const float2 *__restrict__ brightness, const float *__restrict__ vrec, int64_t beam_lw,
    int64_t beam_mh, int64_t beam_nud, const double *__restrict__ lm_ext, const double *__restrict__ freq_data,
    const float *__restrict__ parangles, const float *__restrict__ point_errors, const float *__restrict__ antenna_scaling,
    const float2 *__restrict__ feed_rot, int nsrc, int64_t nchan, int64_t ntime, int nant, int nap, float2 *__restrict__ out,
    int64_t f0, const SuperTileListS tiles)
{
    static_assert(ST % 2 == 0, "sources are consumed in pairs");
    constexpr bool NEG = FORM == FORM_ROW;
    static_assert(!NEG || (RECT && NBR == 8), "the ROW form serves RECT super-tiles of eight block rows");
    using G = GeoS<RECT, NBR, NBC, ST, FORM>;
    constexpr int NAC = G::NAC, CSG = G::CSG, CSH = G::CSH, SRC = G::SRC, BUF = G::BUF, TPS = G::TPS, BT = G::BT;
    extern __shared__ double lds_raw[];
    // doubles first (alignment): per-slot constants, then the phasor table, then the float panels
    double *ldsA = lds_raw;                                    // six planes of per-slot coefficients
    double *ldsU = ldsA + 6 * TPS;                             // (u, v, w) FT per slot, three planes (+ one of padding)
    double2 *ldsT = reinterpret_cast<double2 *>(ldsU + 4 * TPS);
    float2 *ldsR = reinterpret_cast<float2 *>(ldsT + PH_TABLE);
    float *ldsp = reinterpret_cast<float *>(ldsR + 4 * TPS);   // two panel buffers
    const int tid = threadIdx.x;
    const int ptid = tid - S_MATRIX;
    const int64_t f = f0 + blockIdx.y;
    const int t = blockIdx.x;
    const SuperTileS tile = tiles.e[blockIdx.z];
    // antenna of sampling slot a: DIAG -- the super-block's antennas; RECT -- the column super-block's, then the row one's
    auto slot_ant = [&](int a) { return RECT ? (a < NAC ? tile.col_ant0 + a : tile.row_ant0 + a - NAC) : tile.row_ant0 + a; };
    auto slot_ok = [&](int a) {
        const int ant = slot_ant(a);
        return ant < nant && (!RECT || a >= NAC || a < 8 * tile.nc_act);
    };

    fine_table_init(ldsT, tid, S_THREADS);
    const double FT = f4[f] * (PH_TABLE / 4.0);
    const BeamGrid<double> bg = beam_grid<double>(lm_ext, beam_lw, beam_mh, beam_nud);
    const double fscale = freq_data[3 * f + 0];
    for (int a = tid; a < TPS; a += S_THREADS) {
        double sp = 0.0, cp = 1.0, pl = 0.0, pm = 0.0, sl_ = 1.0, sm_ = 1.0, u = 0.0, v = 0.0, w = 0.0;
        if (slot_ok(a)) {
            const int ant = slot_ant(a);
            sincos((double)parangles[(int64_t)t * nant + ant], &sp, &cp);
            const float *pe = point_errors + (((int64_t)t * nant + ant) * nchan + f) * 2;
            const float *as = antenna_scaling + ((int64_t)ant * nchan + f) * 2;
            pl = (double)pe[0]; pm = (double)pe[1]; sl_ = (double)as[0]; sm_ = (double)as[1];
            const double *x = ant_uvw + ((int64_t)t * nant + ant) * 3;
            u = __dmul_rn(x[0], FT); v = __dmul_rn(x[1], FT); w = __dmul_rn(x[2], FT);
        }
        // the slot's coordinate map, folded (fused_voxels_folded): one plane per coefficient
        const double kl = sl_ * bg.lscale, km = sm_ * bg.mscale;
        ldsA[0 * TPS + a] = fscale * cp * kl;
        ldsA[1 * TPS + a] = -fscale * sp * kl;
        ldsA[2 * TPS + a] = (pl * cp - pm * sp) * kl - bg.lower_l * bg.lscale;
        ldsA[3 * TPS + a] = fscale * sp * km;
        ldsA[4 * TPS + a] = fscale * cp * km;
        ldsA[5 * TPS + a] = (pl * sp + pm * cp) * km - bg.lower_m * bg.mscale;
        ldsU[0 * TPS + a] = u; ldsU[1 * TPS + a] = v; ldsU[2 * TPS + a] = w;
    }
    if constexpr (FEED)
        for (int i = tid; i < 4 * TPS; i += S_THREADS)
            ldsR[i] = slot_ok(i >> 2) ? feed_rot[((int64_t)t * nant + slot_ant(i >> 2)) * 4 + (i & 3)] : make_float2(0.0f, 0.0f);
    __syncthreads();
    const int nbatch = (nsrc + ST - 1) / ST;

    if (tid < S_MATRIX) {
        const int32_t *rm = rowmap + (int64_t)t * nap * nap;
        const int lane = tid & 63;
        if constexpr (NEG) {
            switch (tid >> 6) {
            case 0: matrix_wave_row<NBC, ST, 0>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
            case 1: matrix_wave_row<NBC, ST, 1>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
            case 2: matrix_wave_row<NBC, ST, 2>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
            case 3: matrix_wave_row<NBC, ST, 3>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
            case 4: matrix_wave_row<NBC, ST, 4>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
            case 5: matrix_wave_row<NBC, ST, 5>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
            case 6: matrix_wave_row<NBC, ST, 6>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
            default: matrix_wave_row<NBC, ST, 7>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
            }
            return;
        }
        switch (tid >> 6) {
        case 0: matrix_wave_s<RECT, NBR, NBC, ST, FORM, 0>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
        case 1: matrix_wave_s<RECT, NBR, NBC, ST, FORM, 1>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
        case 2: matrix_wave_s<RECT, NBR, NBC, ST, FORM, 2>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
        case 3: matrix_wave_s<RECT, NBR, NBC, ST, FORM, 3>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
        case 4: matrix_wave_s<RECT, NBR, NBC, ST, FORM, 4>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
        case 5: matrix_wave_s<RECT, NBR, NBC, ST, FORM, 5>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
        case 6: matrix_wave_s<RECT, NBR, NBC, ST, FORM, 6>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
        default: matrix_wave_s<RECT, NBR, NBC, ST, FORM, 7>(ldsp, nbatch, lane, rm, nap, tile, nchan, f, out); break;
        }
        return;
    }

    // =================================== sampling waves (8-15) ======================================
    FusedGrid grid;
    {
        grid.lower_l = grid.lower_m = grid.lscale = grid.mscale = 0.0;      // folded into the slots' coefficients
        grid.lmaxf = wave_uniform(bg.lmaxf); grid.mmaxf = wave_uniform(bg.mmaxf);
        grid.lmaxi = __builtin_amdgcn_readfirstlane((int)bg.lmaxi); grid.mmaxi = __builtin_amdgcn_readfirstlane((int)bg.mmaxi);
        grid.stride_m = VREC32 * 4u;
        grid.stride_l = (unsigned)beam_mh * grid.stride_m;
    }
    const int e_corr = ptid & 3;
    const int ei = e_corr >> 1, ej = e_corr & 1;
    const char *plane = reinterpret_cast<const char *>(vrec + (int64_t)blockIdx.y * beam_lw * beam_mh * VREC32);
    const unsigned corr_off = e_corr * 16u;
    __builtin_amdgcn_s_setprio(3);
    struct Term {
        int buf, e_sl, slot, src;
    };
    // batch b, super-round starting at term task0 of the batch; lanes beyond the batch's terms redo an earlier term (same
    // values to the same place), padded antennas and sources beyond the last write zeros
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
    struct Coords {
        double2 lm;
        double n;
    };
    auto fetch = [&](const Term &T) {
        const bool have = T.src < nsrc && slot_ok(T.slot);
        int sidx = have ? T.src : 0;
        if constexpr (TPS == 64) sidx = __builtin_amdgcn_readfirstlane(sidx);
        const double *sp = lmn + 4 * sidx;
        Coords c;
        c.lm = *reinterpret_cast<const double2 *>(sp);
        c.n = sp[2];
        return c;
    };
    Coords nxt = fetch(term_at(0, 0));
    struct Own {                 // a lane's own term of one super-round
        unsigned base, dl, dm;
        float ld, md;
        C2f kph;
        int info;                // LDS offset (floats, panel buffer included) of the term's planes | col_term << 30 | have << 31
        int slot;
        float2 xw[4];            // TPS == 64: the wave's source's brightness (scalar registers)
        float2 xb0, xb1;         // otherwise: column ej of the QUAD's source's brightness
    };
    struct Round {
        F3 v[4];
    };
    auto geometry = [&](const Term &T, const Term &Tn, Own &S) {
        const int e_sl = T.e_sl;
        int e_slot = T.slot;
        asm volatile("" : "+v"(e_slot));
        const bool have = T.src < nsrc && slot_ok(e_slot);
        const bool col_term = !RECT || e_slot < NAC;
        // operand element of the term's antenna (antenna_slot_s: the quads of one store are consecutive elements -- 16
        // consecutive banks per Jones column, the other column 16 banks further, plane_stride); the slots of an absent last
        // column block (RECT, nc_act < NBC) keep their own places behind the real ones
        const int h_slot = !RECT ? antenna_slot_s(e_slot, NAC / 4)
                                 : (!col_term ? 0 : (e_slot < 8 * tile.nc_act ? antenna_slot_s(e_slot, 2 * tile.nc_act) : e_slot));
        const int h_off = T.buf * BUF + e_sl * SRC + 2 * h_slot;
        const int g_off = T.buf * BUF + e_sl * SRC + G::HP * CSH +
                          2 * antenna_slot_s(RECT ? (col_term ? 0 : e_slot - NAC) : e_slot, G::NAR / 4);
        S.info = (RECT ? (col_term ? h_off : g_off) : h_off) | (int)((unsigned)col_term << 30) | (int)((unsigned)have << 31);
        S.slot = e_slot;
        const int bsrc = T.src < nsrc ? T.src : 0;
        if constexpr (TPS == 64) {
            const int us = __builtin_amdgcn_readfirstlane(bsrc);
            const float2 *bp = brightness + ((int64_t)us * nchan + f) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) S.xw[c] = bp[c];
        } else {
            // the four terms of a quad are four consecutive slots of one source (TPS and a super-round's first term are
            // multiples of 4): the brightness column of all four rounds, loaded once and ahead of the gathers
            const float2 *bp = brightness + ((int64_t)bsrc * nchan + f) * 4;
            S.xb0 = bp[ej]; S.xb1 = bp[2 + ej];
        }
        const double2 lm2 = nxt.lm;
        const double nn = nxt.n;
        nxt = fetch(Tn);
        FusedVoxelsC gx;
        fused_voxels_folded(grid, lm2.x, lm2.y, ldsA[0 * TPS + e_slot], ldsA[1 * TPS + e_slot], ldsA[2 * TPS + e_slot],
                            ldsA[3 * TPS + e_slot], ldsA[4 * TPS + e_slot], ldsA[5 * TPS + e_slot], gx);
        S.base = gx.base; S.dl = gx.dl; S.dm = gx.dm;
        S.ld = (float)gx.ld; S.md = (float)gx.md;
        const C2 k = table_phasor(ldsT, fma(nn, ldsU[2 * TPS + e_slot],
                                            fma(lm2.y, ldsU[1 * TPS + e_slot], __dmul_rn(lm2.x, ldsU[0 * TPS + e_slot]))));
        S.kph.re = (float)k.re; S.kph.im = (float)k.im;
    };
    auto issue = [&](auto lane_c, const Own &S, Round &R) {
        constexpr int QL = decltype(lane_c)::value;
        const unsigned base = (unsigned)quad_bcast<QL>((int)S.base) + corr_off;
        const unsigned dl = (unsigned)quad_bcast<QL>((int)S.dl), dm = (unsigned)quad_bcast<QL>((int)S.dm);
        const unsigned offs[4] = {base, base + dl, base + dm, base + dl + dm};
#pragma unroll
        for (int k = 0; k < 4; ++k) R.v[k] = *reinterpret_cast<const F3 *>(plane + (size_t)offs[k]);
    };
    auto finish = [&](auto lane_c, const Own &S, const Round &R) {
        constexpr int QL = decltype(lane_c)::value;
        const int info = quad_bcast<QL>(S.info);
        const bool r_have = info < 0;
        const float ld = quad_bcastf<QL>(S.ld), md = quad_bcastf<QL>(S.md);
        const float omld = __fsub_rn(1.0f, ld), ommd = __fsub_rn(1.0f, md);
        const float wt[4] = {__fmul_rn(omld, ommd), __fmul_rn(ld, ommd), __fmul_rn(omld, md), __fmul_rn(ld, md)};
        C2f kk;
        kk.re = quad_bcastf<QL>(S.kph.re); kk.im = quad_bcastf<QL>(S.kph.im);
        C2f e = beam_reduce1f(R.v, wt);
        if (!r_have) { e.re = 0.0f; e.im = 0.0f; }
        if constexpr (FEED) {
            C2f E0, E1;
            E0.re = pair_bcastf<0>(e.re); E0.im = pair_bcastf<0>(e.im);
            E1.re = pair_bcastf<1>(e.re); E1.im = pair_bcastf<1>(e.im);
            const int r_slot = quad_bcast<QL>(S.slot);
            const float2 r0 = ldsR[4 * r_slot + ej], r1 = ldsR[4 * r_slot + 2 + ej];
            C2f Q0, Q1;
            Q0.re = r0.x; Q0.im = r0.y; Q1.re = r1.x; Q1.im = r1.y;
            e = cmulf(E0, Q0);
            cmacf(e, E1, Q1);
        }
        const C2f A = cmulf(kk, e);
        C2f A0, A1, B0, B1;
        A0.re = pair_bcastf<0>(A.re); A0.im = pair_bcastf<0>(A.im);
        A1.re = pair_bcastf<1>(A.re); A1.im = pair_bcastf<1>(A.im);
        if constexpr (TPS == 64) {
            const float2 b0 = ej ? S.xw[1] : S.xw[0], b1 = ej ? S.xw[3] : S.xw[2];
            B0.re = b0.x; B0.im = b0.y; B1.re = b1.x; B1.im = b1.y;
        } else {
            B0.re = S.xb0.x; B0.im = S.xb0.y; B1.re = S.xb1.x; B1.im = S.xb1.y;
        }
        C2f Gv = cmulf(A0, B0);
        cmacf(Gv, A1, B1);
        if constexpr (!RECT) {
            float *hs = ldsp + (info & 0x3fffffff) + ei;
            float *gs = hs + G::HP * CSH;
            hs[ej * CSH] = A.re; hs[(2 + ej) * CSH] = A.im;
            gs[ej * CSG] = Gv.re; gs[(2 + ej) * CSG] = Gv.im;
if constexpr (FORM == FORM_3M) {
                hs[(4 + ej) * CSH] = __fsub_rn(A.re, A.im); gs[(4 + ej) * CSG] = __fadd_rn(Gv.re, Gv.im);
            }
        } else {
            // a column antenna's term writes its H planes, a row antenna's its G planes
            const bool r_col = (info >> 30) & 1;
            float *ws = ldsp + (info & 0x3fffffff) + ei;
            const int cs = r_col ? CSH : CSG;
            ws[ej * cs] = r_col ? A.re : Gv.re;
            ws[(2 + ej) * cs] = r_col ? A.im : Gv.im;
            if constexpr (FORM != FORM_4P) {
                ws[(4 + ej) * cs] = r_col ? __fsub_rn(A.re, A.im) : __fadd_rn(Gv.re, Gv.im);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // the samplers' side of a batch barrier: their panel writes (LDS) must have landed; the coordinate prefetch (vmcnt)
    // stays outstanding
    auto sampler_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#ifdef AF_C64_STAGE_MATRIX            // (diagnostic build: the matrix waves alone on whatever the LDS holds)
    for (int bb = 0; bb < nbatch; ++bb) sampler_barrier();
    return;
#endif
    {
        Round R0, R1, R2, R3;
        Own S;
        int b = 0, task0 = 0;
        auto next_of = [&](int bb, int tt, int &nb, int &nt) {
            nt = tt + S_SAMPLERS; nb = bb;
            if (nt >= BT) { nt = 0; nb = bb + 1; }
        };
        int b1, t1, b2, t2;
        next_of(b, task0, b1, t1);
        geometry(term_at(b, task0), term_at(b1, t1), S);
        issue(I0{}, S, R0); issue(I1{}, S, R1); issue(I2{}, S, R2); issue(I3{}, S, R3);
        while (true) {
            finish(I0{}, S, R0); finish(I1{}, S, R1); finish(I2{}, S, R2); finish(I3{}, S, R3);
            next_of(b, task0, b1, t1);
            const bool bar = b1 != b, more = b1 < nbatch;
            // (measured and removed, same box, 54.5 ms as it stands: the next super-round's gathers issued BEFORE the batch
            //  barrier and in flight across it 58.4; 16-byte gathers incl. the record's padding word 54.5; without the
            //  sampling waves' raised priority 62.2.  Diagnostic builds -DAF_C64_STAGE_SAMPLE / _MATRIX: the sampling waves
            //  alone 35.8 ms, the matrix waves' 72 000 instructions per workgroup 29.8 ms at one per 32 cycles and SIMD: the
            //  two overlap to 54.5 of their sum of 65.6 -- float32 VALU beside a stream of f32 MFMAs on the same SIMD runs at
            //  ~0.4 of its rate, tools/probe/probe_mfma_f32_16x16x4.hip)
            if (bar) sampler_barrier();
            if (!more) break;
            next_of(b1, t1, b2, t2);
            geometry(term_at(b1, t1), term_at(b2, t2), S);
            issue(I0{}, S, R0); issue(I1{}, S, R1); issue(I2{}, S, R2); issue(I3{}, S, R3);
            b = b1; task0 = t1;
        }
    }
}

struct GemmTilingS {
    int nsb;
    int size[8], blk0[8];
};
constexpr int RECT_COLS_S = 4;
void gemm_tiling_s(int nb, GemmTilingS &tl)
{
    tl.nsb = (nb + 7) / 8;
    for (int i = 0; i < tl.nsb; ++i) {
        tl.size[i] = nb - 8 * i < 8 ? nb - 8 * i : 8;
        tl.blk0[i] = 8 * i;
    }
}

}  // namespace

AF_EXPORT size_t af_fused_predict_c64_workspace_bytes(int64_t nsrc, int64_t nchan, int64_t beam_lw, int64_t beam_mh,
                                                      int64_t beam_nud)
{
    if (nsrc < 0 || nchan < 0 || beam_lw < 0 || beam_mh < 0 || beam_nud < 0) return 0;
    return ws_s(nsrc, nchan, beam_lw, beam_mh, beam_nud).total;
}

// The single-precision form of af_fused_predict_antennas_c128: the plan's arrays as there (ant_uvw in double: the planner
// solves in double whatever the rows' precision), every other array float32 / complex64 (pairs of floats), out complex64.
AF_EXPORT int af_fused_predict_antennas_c64(const double *ant_uvw, const int32_t *rowmap, int64_t nsteps, int64_t nrow,
                                            const float *lm, const float *frequency, const float *brightness, int64_t nsrc,
                                            int64_t nchan, const float *beam, int64_t beam_lw, int64_t beam_mh,
                                            int64_t beam_nud, const float *beam_lm_extents, const float *beam_freq_map,
                                            const float *parallactic_angles, int64_t ntime, int64_t nant,
                                            const float *point_errors, const float *antenna_scaling,
                                            const float *feed_rotation, int convention, float *out, void *workspace,
                                            size_t workspace_bytes, void *stream)
{
    AF_REQUIRE(convention == AF_CONVENTION_FOURIER || convention == AF_CONVENTION_CASA,
               "convention not in ('fourier', 'casa')");
    AF_REQUIRE(beam_lw >= 2 && beam_mh >= 2 && beam_nud >= 2, "beam_lw, beam_mh and beam_nud must be >= 2");
    AF_REQUIRE(nsteps >= 0 && nrow >= 0 && nsrc >= 0 && nchan >= 0 && ntime >= 0 && nant >= 0,
               "af_fused_predict_antennas_c64: negative extent");
    AF_REQUIRE(nant <= 512, "af_fused_predict_antennas_c64: more than 512 antennas");
    AF_REQUIRE(nsteps <= ntime, "af_fused_predict_antennas_c64: %lld steps but %lld timesteps of per-antenna terms",
               (long long)nsteps, (long long)ntime);
    AF_REQUIRE(nsrc < (1LL << 31) && nchan <= 65535 && nsteps < (1LL << 31), "af_fused_predict_antennas_c64: too large");
    hipStream_t st_ = af_stream(stream);
    if (nrow == 0 || nchan == 0 || nsteps == 0) return AF_OK;
    AF_REQUIRE(out != nullptr, "af_fused_predict_antennas_c64: out is NULL");
    if (nsrc == 0) {
        AF_HIP(hipMemsetAsync(out, 0, sizeof(float) * 2 * 4 * (size_t)(nrow * nchan), st_));
        return AF_OK;
    }
    AF_REQUIRE(ant_uvw && rowmap && lm && frequency && brightness && beam && beam_lm_extents && beam_freq_map &&
                   parallactic_angles && point_errors && antenna_scaling,
               "af_fused_predict_antennas_c64: NULL array");
    const WsS W = ws_s(nsrc, nchan, beam_lw, beam_mh, beam_nud);
    AF_REQUIRE(workspace != nullptr && workspace_bytes >= W.total,
               "af_fused_predict_antennas_c64: workspace too small (%zu < %zu)", workspace_bytes, W.total);
    AF_REQUIRE(((uintptr_t)workspace & 255) == 0, "af_fused_predict_antennas_c64: workspace must be 256-byte aligned");
    char *ws = static_cast<char *>(workspace);
    double *lmn = reinterpret_cast<double *>(ws + W.lmn), *f4 = reinterpret_cast<double *>(ws + W.f4);
    double *freq_d = reinterpret_cast<double *>(ws + W.freq_d), *fmap_d = reinterpret_cast<double *>(ws + W.fmap_d);
    double *ext_d = reinterpret_cast<double *>(ws + W.ext_d), *freq_data = reinterpret_cast<double *>(ws + W.freq_data);
    float *planes_buf = reinterpret_cast<float *>(ws + W.planes);
    hipLaunchKernelGGL(prep_src_f32, dim3((unsigned)af_cdiv(nsrc, 256)), dim3(256), 0, st_, lm, nsrc, lmn);
    AF_LAUNCH_CHECK();
    const int64_t nprep = nchan > beam_nud ? (nchan > 4 ? nchan : 4) : (beam_nud > 4 ? beam_nud : 4);
    hipLaunchKernelGGL(prep_freq_f32, dim3((unsigned)af_cdiv(nprep, 64)), dim3(64), 0, st_, frequency, nchan, convention,
                       beam_freq_map, beam_nud, beam_lm_extents, f4, freq_d, fmap_d, ext_d);
    AF_LAUNCH_CHECK();
    int rc = af_freq_grid_interp_f64(freq_d, nchan, fmap_d, beam_nud, freq_data, stream);
    if (rc != AF_OK) return rc;
    const int64_t ncell = beam_lw * beam_mh;
    AF_REQUIRE(ncell < (1LL << 25), "af_fused_predict_antennas_c64: beam cube too large (fewer than 2^25 cells per plane)");
    const bool feed = feed_rotation != nullptr;
    const int nb = (int)((nant + 7) / 8), nap = 8 * nb;
    GemmTilingS tl;
    gemm_tiling_s(nb, tl);
    struct Shape {
        const void *kernel;
        size_t lds;
        SuperTileListS list;
        int count;
    };
    std::vector<Shape> shapes;
    auto add = [&](const void *kernel, size_t lds, const SuperTileS &e) {
        for (auto &s : shapes)
            if (s.kernel == kernel && s.count < 16) { s.list.e[s.count++] = e; return; }
        Shape s;
        memset(&s, 0, sizeof(s));
        s.kernel = kernel; s.lds = lds; s.list.e[0] = e; s.count = 1;
        shapes.push_back(s);
    };
#define AF_GEMMS_K(RECTC, NBRC, NBCC, STC)                                                                             \
    (feed ? reinterpret_cast<const void *>(fused_gemm_c64_kernel<true, RECTC, NBRC, NBCC, STC, RECTC ? FORM_4P : FORM_3M>)  \
          : reinterpret_cast<const void *>(fused_gemm_c64_kernel<false, RECTC, NBRC, NBCC, STC, RECTC ? FORM_4P : FORM_3M>)), \
        GeoS<RECTC, NBRC, NBCC, STC, RECTC ? FORM_4P : FORM_3M>::lds_bytes()
#define AF_GEMMS_KN(NBCC, STC)                                                                                         \
    (feed ? reinterpret_cast<const void *>(fused_gemm_c64_kernel<true, true, 8, NBCC, STC, FORM_ROW>)                   \
          : reinterpret_cast<const void *>(fused_gemm_c64_kernel<false, true, 8, NBCC, STC, FORM_ROW>)),                \
        GeoS<true, 8, NBCC, STC, FORM_ROW>::lds_bytes()
    for (int i = 0; i < tl.nsb; ++i) {
        SuperTileS e = {8 * tl.blk0[i], 8 * tl.blk0[i], tl.size[i], 0};
        switch (tl.size[i]) {
        // sources per batch: one super-round of the 512 sampling lanes where two buffers of six + six planes fit the LDS
        case 1: add(AF_GEMMS_K(false, 1, 1, 16), e); break;
        case 2: add(AF_GEMMS_K(false, 2, 2, 24), e); break;
        case 3: add(AF_GEMMS_K(false, 3, 3, 16), e); break;
        case 4: add(AF_GEMMS_K(false, 4, 4, 16), e); break;
        case 5: add(AF_GEMMS_K(false, 5, 5, 12), e); break;
        case 6: add(AF_GEMMS_K(false, 6, 6, 10), e); break;
        case 7: add(AF_GEMMS_K(false, 7, 7, 8), e); break;
        default: add(AF_GEMMS_K(false, 8, 8, 8), e); break;
        }
        for (int j = i + 1; j < tl.nsb; ++j) {
            if (tl.size[j] > RECT_COLS_S) {        // one 8 x 8 super-tile (ROW form) per pair of super-blocks
                SuperTileS r = {8 * tl.blk0[i], 8 * tl.blk0[j], tl.size[j], 0};
                add(AF_GEMMS_KN(8, 8), r);
                continue;
            }
            for (int c0 = 0; c0 < tl.size[j]; c0 += RECT_COLS_S) {
                SuperTileS r = {8 * tl.blk0[i], 8 * (tl.blk0[j] + c0), tl.size[j] - c0 < RECT_COLS_S ? tl.size[j] - c0 : RECT_COLS_S, 0};
                add(AF_GEMMS_K(true, 8, RECT_COLS_S, 16), r);
            }
        }
    }
#undef AF_GEMMS_K
#undef AF_GEMMS_KN
    for (auto &s : shapes) {
        AF_REQUIRE(s.lds <= 160 * 1024, "af_fused_predict_antennas_c64: %zu bytes of LDS needed", s.lds);
        AF_HIP(hipFuncSetAttribute(s.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lds));
    }
    int nsrc_i = (int)nsrc, nant_i = (int)nant, nap_i = nap;
    const float2 *b2 = reinterpret_cast<const float2 *>(brightness), *fr2 = reinterpret_cast<const float2 *>(feed_rotation);
    float2 *out2 = reinterpret_cast<float2 *>(out);
    const double *ext_c = ext_d;
    for (int64_t f0 = 0; f0 < nchan; f0 += PLANE_GROUP) {
        const int64_t nf = nchan - f0 < PLANE_GROUP ? nchan - f0 : PLANE_GROUP;
        int64_t blocks = af_cdiv(ncell * 4, 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(beam_plane_kernel_f32, dim3((unsigned)blocks, (unsigned)nf), dim3(256), 0, st_,
                           reinterpret_cast<const float2 *>(beam), ncell, beam_nud, freq_data, f0, planes_buf);
        AF_LAUNCH_CHECK();
        if (f0 == 0) af_prof_begin(st_);
        for (auto &s : shapes) {
            void *args[] = {&ant_uvw, &rowmap, &lmn, &f4, &b2, &planes_buf, &beam_lw, &beam_mh, &beam_nud, &ext_c,
                            &freq_data, &parallactic_angles, &point_errors, &antenna_scaling, &fr2, &nsrc_i, &nchan, &ntime,
                            &nant_i, &nap_i, &out2, &f0, &s.list};
            AF_HIP(hipLaunchKernel(s.kernel, dim3((unsigned)nsteps, (unsigned)nf, (unsigned)s.count), dim3(S_THREADS), args,
                                   s.lds, st_));
        }
        if (f0 == 0) af_prof_end(st_);
        AF_LAUNCH_CHECK();
    }
    return AF_OK;
}
