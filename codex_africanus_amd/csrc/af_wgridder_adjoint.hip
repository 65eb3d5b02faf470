// ================= the adjoint: visibilities -> image (africanus/gridding/wgridder/vis2im.py:15-72, ducc0's ms2dirty) =====
//     dirty[x, y] = (1 / n) sum_{r, c} Re( wgt vis exp(+2 pi i nu/c (u x + v y - w (n - 1))) )
// (test_wgridder.py:18-46).  Exactly the transpose of the operator above, plane by plane: the visibilities are spread
// with the same taps onto the w-plane grids, every grid is transformed back (rows along u, the nx image rows gathered
// into the staging array, rows along v), multiplied by exp(-2 pi i w_k (n - 1)) and by the same taper A, and its real
// part added to the image.
//
// NOT ON THE SURVEY 8 HOT PATH: this translation unit is the ADJOINT side only (wg_grid_planes, wg_grid_tiles,
// wg_gather_rows, wg_add_plane; entry af_wgrid_vis2im_f64 in af_wgridder.hip, whose host section sorts the visibilities,
// runs the plane transforms and calls the three launchers at the bottom of this file).  Built in round 2 beside the forward
// operator because the reference's only pin for its wgridder wrappers is the pair's adjointness and DFT accuracy
// (gridding/wgridder/tests/test_wgridder.py:116-); kept as the forward path's cross-check, not tuned since, not a roofline
// row.  The forward path (BASELINE configs[4] as named) uses nothing of this file.
#include <type_traits>

#include "af_common.h"
#include "af_wgrid_device.h"
#include "af_wgrid_taps.h"

namespace {

// (small calls) one lane per visibility, hardware fp64 atomics into the planes
template <int W>
__global__ __launch_bounds__(256) void wg_grid_planes(const double *__restrict__ uvw, const double *__restrict__ freq,
                                                      int64_t nrow, int64_t nchan_b, int64_t chan0, int64_t nchan_total,
                                                      double2 *__restrict__ grids, int64_t nu, int64_t nv, double cellx,
                                                      double celly, double beta, double w0, double dw, int pk0, int pk1,
                                                      int do_w, const unsigned char *__restrict__ mask,
                                                      const double *__restrict__ wgt, const double2 *__restrict__ vis,
                                                      const WgPoly poly)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrow * nchan_b) return;
    const int64_t r = i / nchan_b, c = i - r * nchan_b;
    const int64_t o = r * nchan_total + chan0 + c;
    if (mask && !mask[o]) return;
    double2 val = vis[o];
    if (wgt) { val.x *= wgt[o]; val.y *= wgt[o]; }
    const double sg = wg_fold_sign(uvw + 3 * r, do_w);
    val.y *= sg;                                            // w < 0: conj(vis) at the mirrored point
    const double fl = sg * (freq[c] / AF_LIGHTSPEED);
    double gw = 0.0;
    int k0 = 0, k1 = 1, k0u = 0;
    if (do_w) {
        gw = (uvw[3 * r + 2] * fl - w0) / dw;
        if (!isfinite(gw)) return;
        k0 = (int)ceil(gw - 0.5 * W);
        k1 = k0 + W;
        k0u = k0;
        k0 = k0 < pk0 ? pk0 : k0;
        k1 = k1 > pk1 ? pk1 : k1;
        if (k0 >= k1) return;
    }
    const double gu = uvw[3 * r + WG_CU] * fl * cellx * (double)nu, gv = uvw[3 * r + WG_CV] * fl * celly * (double)nv;
    if (!(isfinite(gu) && isfinite(gv) && fabs(gu) < 1e15 && fabs(gv) < 1e15)) return;
    const double fu = ceil(gu - 0.5 * W) - gu, fv = ceil(gv - 0.5 * W) - gv;
    const int pu0 = wg_first_cell(gu, W, (int)nu), pv0 = wg_first_cell(gv, W, (int)nv);
    double ku[W], kv[W], kwv[W];
#pragma unroll
    for (int t = 0; t < W; ++t) kwv[t] = t == 0 ? 1.0 : 0.0;
    wg_taps<W>(poly, fu, beta, ku);
    wg_taps<W>(poly, fv, beta, kv);
    if (do_w) wg_taps<W>(poly, (double)k0u - gw, beta, kwv);
    for (int k = k0; k < k1; ++k) {
        const double kw = wg_pick<W>(kwv, k - k0u);
        double *__restrict__ grid = reinterpret_cast<double *>(grids + (int64_t)(k - pk0) * nu * nv);
#pragma unroll
        for (int a = 0; a < W; ++a) {
            int pa = pu0 + a;
            pa = pa >= nu ? pa - (int)nu : pa;
            const double wa = kw * ku[a];
#pragma unroll
            for (int b = 0; b < W; ++b) {
                int pb = pv0 + b;
                pb = pb >= nv ? pb - (int)nv : pb;
                const double wt = wa * kv[b];
                double *cell = grid + 2 * ((int64_t)pa * nv + pb);
                unsafeAtomicAdd(cell, wt * val.x);
                unsafeAtomicAdd(cell + 1, wt * val.y);
            }
        }
    }
}

// (large calls) The visibilities sorted by (tile, first w-plane) -- exactly: one sort bucket per plane -- are taken
// through LDS: a workgroup owns a chunk of <= 4096 visibilities of one tile and keeps the tile's cells of W consecutive
// planes in a ring of LDS images (plane k in slot k mod W).  A visibility adds its W x W taps to its W planes with plain
// LDS read-add-writes (one lane per tap: distinct cells; every image belongs to one wave, so no atomics); when the sorted
// list moves on to a higher first plane, the planes that can receive nothing more are added to the grids in memory
// (hardware fp64 atomics: neighbouring tiles share the halo cells) and their slots cleared.  Every (chunk, plane) is
// flushed once: W^3 atomics per visibility become (T + W - 1)^2 per (chunk, plane).
template <int W>
__global__ __launch_bounds__(256) void wg_grid_tiles(const double *__restrict__ uvw, const double *__restrict__ freq,
                                                     int64_t nchan_b, int64_t chan0, int64_t nchan_total,
                                                     double2 *__restrict__ grids, int64_t nu, int64_t nv, double cellx,
                                                     double celly, double beta, double w0, double dw, int pk0, int pk1,
                                                     int do_w, const unsigned *__restrict__ idx, const int *__restrict__ start,
                                                     int kb, const int2 *__restrict__ chunks, const int *__restrict__ nchunks,
                                                     const double *__restrict__ wgt, const double2 *__restrict__ vis,
                                                     const WgPoly poly)
{
    constexpr int T = wg_gtile(W), R = T + W - 1, RR = R * R;
    constexpr int NT = 4 * W;                   // table doubles per visibility: val.re ku[], val.im ku[], kv[], kw[]
    constexpr int NE = (2 * RR + 63) / 64;      // doubles of the region per lane (flush)
    constexpr int NP = (W * W + 63) / 64;       // tap passes (one for W <= 8)
    __shared__ double2 ring[W * RR];
    __shared__ double tab[64 * NT];
    if ((int)blockIdx.x >= *nchunks) return;
    const int2 ch = chunks[blockIdx.x];
    // four waves share the chunk: every wave walks ALL its visibilities, but adds only to the ring slots it owns
    // (slot % 4 == wave) -- no two waves ever touch the same LDS image, so no atomics and no barriers in the walk
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nty = (int)((nv + T - 1) / T);
    const int tu = ch.x / nty, tv = ch.x - tu * nty;
    int n = start[(ch.x + 1) * kb] - ch.y;
    n = n > WG_GCHUNK ? WG_GCHUNK : n;
    const int64_t plane = nu * nv;

    for (int e = tid; e < W * RR; e += 256) ring[e] = make_double2(0.0, 0.0);
    // The flush works on DOUBLES, not cells: lane l of pass q takes double l + 64 q of the region's 2 R R (re, im
    // interleaved), so that one atomic instruction covers whole contiguous runs of a grid row -- 64 consecutive doubles =
    // four full 128-byte lines -- instead of every other double of twice as many lines (the flush atomics are what bounds
    // this kernel: 4.8e9 of them per call at the ~1.6e11 / s the memory side sustains)
    int gofs[NE];                               // grid offset (in doubles) of this lane's doubles of the region, wrapped
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int d = lane + 64 * q, e = d >> 1, a = e / R, b = e - a * R;
        int gu_ = tu * T + a, gv_ = tv * T + b;          // wrapped by subtraction: no 64-bit %
        while (gu_ >= (int)nu) gu_ -= (int)nu;
        while (gv_ >= (int)nv) gv_ -= (int)nv;
        gofs[q] = d < 2 * RR ? (int)(2 * ((int64_t)gu_ * nv + gv_) + (d & 1)) : -1;
    }
    // this lane's tap(s): row a, column b, offset a R + b -- dealt to the lanes so that the lane groups of the 16-byte
    // LDS accesses repeat as few cells mod 16 (reads) / mod 8 (writes) as possible (af_wgrid_taps.h; in row-major lane
    // order W = 7 paid 11 conflict cycles on top of the 12 of one read + write, and the LDS is what bounds this kernel)
    static_assert(WgTaps<W>::NP == NP, "tap table and kernel disagree on the number of passes");
    int ta[NP], tb[NP], tcell[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int t = WgTaps<W>::tap(p, lane);
        ta[p] = t >= 0 ? t / W : -1;
        tb[p] = t >= 0 ? t % W : 0;
        tcell[p] = t >= 0 ? (t / W) * R + t % W : 0;
    }
    // plane k -> memory, slot cleared (by the wave that owns the slot)
    auto retire = [&](int k) {
        const int slot = ((k % W) + W) % W;
        if ((slot & 3) != wave) return;
        const bool live = k >= pk0 && k < pk1;
        double *__restrict__ g = reinterpret_cast<double *>(grids + (int64_t)(k - pk0) * plane);
        double *__restrict__ rs = reinterpret_cast<double *>(ring + slot * RR);
        double v[NE];                           // all the reads first: one LDS round trip per plane, not NE
#pragma unroll
        for (int q = 0; q < NE; ++q) v[q] = gofs[q] >= 0 ? rs[lane + 64 * q] : 0.0;
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            if (v[q] != 0.0) {
                if (live) unsafeAtomicAdd(g + gofs[q], v[q]);
                rs[lane + 64 * q] = 0.0;
            }
        }
    };
    int kcur = 0;
    bool started = false;
    __syncthreads();
    // inputs of visibilities base .. base + 63 (lane v of every wave: visibility base + v): loaded one batch ahead, so
    // that the two dependent memory round trips (sorted index -> row) hide behind the walk of the batch before
    struct Inputs { double2 val; double u, v, w, fl; bool valid; };
    auto load_inputs = [&](int base) {
        Inputs in;
        in.valid = base + lane < n;
        in.val = make_double2(0.0, 0.0);
        in.u = in.v = in.w = in.fl = 0.0;
        if (in.valid) {
            const unsigned i = idx[ch.y + base + lane];
            const unsigned r = i / (unsigned)nchan_b, c = i - r * (unsigned)nchan_b;
            const int64_t o = (int64_t)r * nchan_total + chan0 + c;
            const double sg = wg_fold_sign(uvw + 3 * (int64_t)r, do_w);
            in.fl = sg * (freq[c] / AF_LIGHTSPEED);
            in.u = uvw[3 * (int64_t)r + WG_CU];
            in.v = uvw[3 * (int64_t)r + WG_CV];
            in.w = uvw[3 * (int64_t)r + 2];
            if (wave == 0) {
                in.val = vis[o];
                if (wgt) { const double g = wgt[o]; in.val.x *= g; in.val.y *= g; }
                in.val.y *= sg;                             // w < 0: conj(vis) at the mirrored point
            }
        }
        return in;
    };
    Inputs nxt = load_inputs(0);
    for (int base = 0; base < n; base += 64) {
        // the table of visibilities base .. base + 63: wave 0 writes val ku[], wave 1 kv[], wave 2 the plane weights;
        // every wave keeps the visibility's first plane and offset
        const Inputs in = nxt;
        int k0 = 0x7fffffff, lofs = 0;
        if (in.valid) {
            const double fl = in.fl;
            double gw = 0.0;
            k0 = 0;
            if (do_w) {
                gw = (in.w * fl - w0) / dw;
                k0 = (int)ceil(gw - 0.5 * W);
            }
            const double gu = in.u * fl * cellx * (double)nu;
            const double gv = in.v * fl * celly * (double)nv;
            double *__restrict__ t = tab + lane * NT;
            double kk[W];
            if (wave == 0) {
                const double2 val = in.val;
                wg_taps<W>(poly, ceil(gu - 0.5 * W) - gu, beta, kk);
#pragma unroll
                for (int a = 0; a < W; ++a) {
                    t[a] = val.x * kk[a];
                    t[W + a] = val.y * kk[a];
                }
            } else if (wave == 1) {
                wg_taps<W>(poly, ceil(gv - 0.5 * W) - gv, beta, kk);
#pragma unroll
                for (int a = 0; a < W; ++a) t[2 * W + a] = kk[a];
            } else if (wave == 2) {
                // the plane weights in SLOT order (plane k0 + a lives in slot (k0 + a) mod W): the walk then reads
                // them at compile-time offsets and needs no per-visibility scalar arithmetic
                int sl = ((k0 % W) + W) % W;
#pragma unroll
                for (int a = 0; a < W; ++a) kk[a] = a == 0 ? 1.0 : 0.0;
                if (do_w) wg_taps<W>(poly, (double)k0 - gw, beta, kk);
#pragma unroll
                for (int a = 0; a < W; ++a) {
                    t[3 * W + sl] = kk[a];
                    sl = sl + 1 == W ? 0 : sl + 1;
                }
            }
            lofs = (wg_first_cell(gu, W, (int)nu) - tu * T) * R + wg_first_cell(gv, W, (int)nv) - tv * T;
        }
        if (base + 64 < n) nxt = load_inputs(base + 64);
        __syncthreads();
        const int nb = n - base < 64 ? n - base : 64;
        // the walk, compiled once per wave number so that the slots a wave owns are compile-time constants (a dynamic
        // ownership test per slot cost ~40 scalar branches per visibility)
        auto walk = [&](auto wvc) {
            constexpr int WV = decltype(wvc)::value;
            constexpr int NS = (W - WV + 3) / 4;      // slots WV, WV + 4, ...
            for (int j = 0; j < nb; ++j) {
                const int k0j = __builtin_amdgcn_readlane(k0, j), lofsj = __builtin_amdgcn_readlane(lofs, j);
                if (!started) { kcur = k0j; started = true; }
                if (k0j > kcur) {                // planes below k0j are complete for this chunk
                    const int upto = k0j - kcur < W ? k0j : kcur + W;
                    for (int k = kcur; k < upto; ++k) retire(k);
                    kcur = k0j;
                }
                if (NS == 0) continue;
                const double *__restrict__ t = tab + j * NT;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (ta[p] < 0) continue;
                    const double kvb = t[2 * W + tb[p]];
                    const double pre = t[ta[p]] * kvb, pim = t[W + ta[p]] * kvb;
                    const int cell = lofsj + tcell[p];
                    double2 v[NS > 0 ? NS : 1];
                    double kw[NS > 0 ? NS : 1];
#pragma unroll
                    for (int m = 0; m < NS; ++m) {
                        kw[m] = t[3 * W + WV + 4 * m];
                        v[m] = ring[(WV + 4 * m) * RR + cell];
                    }
#pragma unroll
                    for (int m = 0; m < NS; ++m) {
                        v[m].x = fma(kw[m], pre, v[m].x);
                        v[m].y = fma(kw[m], pim, v[m].y);
                    }
#pragma unroll
                    for (int m = 0; m < NS; ++m) ring[(WV + 4 * m) * RR + cell] = v[m];
                }
            }
        };
        switch (wave) {
        case 0: walk(std::integral_constant<int, 0>{}); break;
        case 1: walk(std::integral_constant<int, 1>{}); break;
        case 2: walk(std::integral_constant<int, 2>{}); break;
        default: walk(std::integral_constant<int, 3>{}); break;
        }
        __syncthreads();
    }
    if (started)
        for (int k = kcur; k < kcur + W; ++k) retire(k);
}

// S[ix * nv + pv] = G[pv * nu + pu(ix)]: the nx image rows of a plane (transformed along u), v contiguous again
__global__ __launch_bounds__(256) void wg_gather_rows(const double2 *__restrict__ G, int64_t nx, int64_t nu, int64_t nv,
                                                      double2 *__restrict__ S)
{
    __shared__ double2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t ix0 = (int64_t)blockIdx.x * 32, pv0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t pv = pv0 + ty + 8 * j, ix = ix0 + tx;
        int64_t pu = ix - nx / 2;
        pu = pu < 0 ? pu + nu : pu;
        tile[ty + 8 * j][tx] = (pv < nv && ix < nx) ? G[pv * nu + pu] : make_double2(0.0, 0.0);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t ix = ix0 + ty + 8 * j, pv = pv0 + tx;
        if (ix < nx && pv < nv) S[ix * nv + pv] = tile[tx][ty + 8 * j];
    }
}

// image[ix, iy] (+)= A Re( S[ix, pv(iy)] exp(-2 pi i w_k (n - 1)) )
__global__ __launch_bounds__(256) void wg_add_plane(const double2 *__restrict__ S, const double *__restrict__ A,
                                                    const double *__restrict__ nm1, int64_t nx, int64_t ny, int64_t nv,
                                                    double wk, int first, double *__restrict__ image)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx * ny) return;
    const int64_t ix = i / ny, iy = i - ix * ny;
    int64_t pv = iy - ny / 2;
    pv = pv < 0 ? pv + nv : pv;
    const double2 g = S[ix * nv + pv];
    double sn, cs;
    sincospi(2.0 * wk * nm1[i], &sn, &cs);
    const double v = A[i] * (g.x * cs + g.y * sn);
    image[i] = first ? v : image[i] + v;
}

}  // namespace

void wg_adjoint_spread(int kernel_width, bool tiled, unsigned blocks, hipStream_t st, const WgSpreadArgs &a, const WgPoly &poly)
{
#define AF_WG_LAUNCH(WC)                                                                                               \
    if (tiled)                                                                                                         \
        hipLaunchKernelGGL((wg_grid_tiles<WC>), dim3(blocks), dim3(256), 0, st, a.uvw, a.freq, a.nchan_b, a.chan0,       \
                           a.nchan_total, a.grids, a.nu, a.nv, a.cellx, a.celly, a.beta, a.w0, a.dw, a.pk0, a.pk1,       \
                           a.do_w, a.idx, a.start, a.kb, a.chunks, a.nchunks, a.wgt, a.vis, poly);                       \
    else                                                                                                               \
        hipLaunchKernelGGL((wg_grid_planes<WC>), dim3(blocks), dim3(256), 0, st, a.uvw, a.freq, a.nrow, a.nchan_b,       \
                           a.chan0, a.nchan_total, a.grids, a.nu, a.nv, a.cellx, a.celly, a.beta, a.w0, a.dw, a.pk0,     \
                           a.pk1, a.do_w, a.mask, a.wgt, a.vis, poly)
    switch (kernel_width) {
    case 4: AF_WG_LAUNCH(4); break;
    case 5: AF_WG_LAUNCH(5); break;
    case 6: AF_WG_LAUNCH(6); break;
    case 7: AF_WG_LAUNCH(7); break;
    case 8: AF_WG_LAUNCH(8); break;
    case 9: AF_WG_LAUNCH(9); break;
    case 10: AF_WG_LAUNCH(10); break;
    case 11: AF_WG_LAUNCH(11); break;
    case 12: AF_WG_LAUNCH(12); break;
    case 13: AF_WG_LAUNCH(13); break;
    case 14: AF_WG_LAUNCH(14); break;
    case 15: AF_WG_LAUNCH(15); break;
    default: AF_WG_LAUNCH(16); break;
    }
#undef AF_WG_LAUNCH
}

void wg_adjoint_gather_rows(const double2 *G, int64_t nx, int64_t nu, int64_t nv, double2 *S, hipStream_t st)
{
    hipLaunchKernelGGL(wg_gather_rows, dim3((unsigned)af_cdiv(nx, 32), (unsigned)af_cdiv(nv, 32)), dim3(256), 0, st, G, nx, nu,
                       nv, S);
}

void wg_adjoint_add_plane(const double2 *S, const double *A, const double *nm1, int64_t nx, int64_t ny, int64_t nv, double wk,
                          int first, double *image, hipStream_t st)
{
    hipLaunchKernelGGL(wg_add_plane, dim3((unsigned)af_cdiv(nx * ny, 256)), dim3(256), 0, st, S, A, nm1, nx, ny, nv, wk, first,
                       image);
}
