// Quarter-turn polynomial sincos shared by the im_to_vis and fused-predict kernels.
#pragma once
#include "af_common.h"

// (cos, sin)(2*pi*t) for t given in QUARTER turns t4 = 4*t.  Reduction: r = rint(t4) by the
// 1.5*2^52 magic-number add (also yields the quadrant in the low dword), f = t4 - r exact in
// [-0.5, 0.5]; sin(pi/2 f) = f*S(f^2), cos(pi/2 f) = C(f^2), Chebyshev-node fits on
// f^2 in [0, 0.25]: |err| <= 7e-15 (S), 6e-14 (C) with 6 terms; ~1e-16 with 7 terms (and exactly
// (1, 0) at f = 0, so a source at the phase centre gets a unit phasor).  The kernels use 7: an
// error e in cos of the channel-step angle is amplified by the recurrence to ~j^2*e at channel j.
template <int NTERM>
__device__ __forceinline__ void sincos_quarter_turns(double t4, double &c_out, double &s_out)
{
    static_assert(NTERM == 6 || NTERM == 7, "6 or 7 polynomial terms");
    constexpr double S6[7] = {0x1.921fb54442cfap+0, -0x1.4abbce6257a2ap-1, 0x1.466bc67123fa1p-4,
                              -0x1.32d2c644adc0bp-8, 0x1.5071ce4b47930p-13, -0x1.dd54805f3f706p-19, 0.0};
    constexpr double C6[7] = {0x1.ffffffffffe0bp-1, -0x1.3bd3cc9bd2c35p+0, 0x1.03c1f074ded21p-2,
                              -0x1.55d3ba300cd50p-6, 0x1.e1e7ccccb387ap-11, -0x1.a0ee132c60c1fp-16, 0.0};
    constexpr double S7[7] = {0x1.921fb54442d18p+0, -0x1.4abbce625be41p-1, 0x1.466bc677587f8p-4,
                              -0x1.32d2cce2e5b19p-8, 0x1.50782fda12d96p-13, -0x1.e30071afc3e59p-19,
                              0x1.e3f38399551bfp-25};
    constexpr double C7[7] = {0x1.0000000000000p+0, -0x1.3bd3cc9be458bp+0, 0x1.03c1f081b0780p-2,
                              -0x1.55d3c7dbfd139p-6, 0x1.e1f4fb60281f6p-11, -0x1.a6c9c1be9eb49p-16,
                              0x1.f3dbcea61b1a4p-22};
    const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
    double a = __dadd_rn(t4, MAGIC);
    int q = __double2loint(a);                // low 32 bits = rint(t4) mod 2^32
    double r = __dsub_rn(a, MAGIC);
    double f = __dsub_rn(t4, r);
    double z = __dmul_rn(f, f);
    double ps = NTERM == 6 ? S6[5] : S7[6], pc = NTERM == 6 ? C6[5] : C7[6];
#pragma unroll
    for (int i = NTERM - 2; i >= 0; --i) {
        ps = fma(ps, z, NTERM == 6 ? S6[i] : S7[i]);
        pc = fma(pc, z, NTERM == 6 ? C6[i] : C7[i]);
    }
    ps = __dmul_rn(ps, f);
    // quadrant: 0 (c,s)  1 (-s,c)  2 (-c,-s)  3 (s,-c)
    const bool swap = q & 1;
    double cc = swap ? ps : pc;
    double ss = swap ? pc : ps;
    int chi = __double2hiint(cc) ^ (((q + 1) & 2) << 30);
    int shi = __double2hiint(ss) ^ ((q & 2) << 30);
    c_out = __hiloint2double(chi, __double2loint(cc));
    s_out = __hiloint2double(shi, __double2loint(ss));
}


// The same evaluation (7 terms) cut into stages, for kernels that spread one sincos over several
// scheduling regions: reduce -> horner<5,3> -> horner<2,0> -> finish gives bit-identical results to
// sincos_quarter_turns<7>.
struct SinCosStage {
    double f, z, ps, pc;
    int q;
};
namespace af_sincos_detail {
__device__ constexpr double S7[7] = {0x1.921fb54442d18p+0, -0x1.4abbce625be41p-1, 0x1.466bc677587f8p-4,
                                     -0x1.32d2cce2e5b19p-8, 0x1.50782fda12d96p-13, -0x1.e30071afc3e59p-19,
                                     0x1.e3f38399551bfp-25};
__device__ constexpr double C7[7] = {0x1.0000000000000p+0, -0x1.3bd3cc9be458bp+0, 0x1.03c1f081b0780p-2,
                                     -0x1.55d3c7dbfd139p-6, 0x1.e1f4fb60281f6p-11, -0x1.a6c9c1be9eb49p-16,
                                     0x1.f3dbcea61b1a4p-22};
}  // namespace af_sincos_detail
__device__ __forceinline__ void sincos_qt_reduce(SinCosStage &s, double t4)
{
    const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
    const double a = __dadd_rn(t4, MAGIC);
    s.q = __double2loint(a);
    const double r = __dsub_rn(a, MAGIC);
    s.f = __dsub_rn(t4, r);
    s.z = __dmul_rn(s.f, s.f);
    s.ps = af_sincos_detail::S7[6];
    s.pc = af_sincos_detail::C7[6];
}
template <int HI, int LO>
__device__ __forceinline__ void sincos_qt_horner(SinCosStage &s)
{
#pragma unroll
    for (int i = HI; i >= LO; --i) {
        s.ps = fma(s.ps, s.z, af_sincos_detail::S7[i]);
        s.pc = fma(s.pc, s.z, af_sincos_detail::C7[i]);
    }
}
__device__ __forceinline__ void sincos_qt_finish(const SinCosStage &s, double &c_out, double &s_out)
{
    const double ps = __dmul_rn(s.ps, s.f);
    const bool swap = s.q & 1;
    const double cc = swap ? ps : s.pc;
    const double ss = swap ? s.pc : ps;
    const int chi = __double2hiint(cc) ^ (((s.q + 1) & 2) << 30);
    const int shi = __double2hiint(ss) ^ ((s.q & 2) << 30);
    c_out = __hiloint2double(chi, __double2loint(cc));
    s_out = __hiloint2double(shi, __double2loint(ss));
}


// exp(2 pi i x / 256) from a 256-entry table of exp(2 pi i k / 256) (LDS: one 16-byte gather) times the residual
// rotation |theta| <= pi/256 by its Taylor series -- sin to theta^5, cos to theta^6: truncation 8e-18 / 1e-20; the
// result carries the table entry's and four products' roundings (~2e-16).  x in units of 1/256 turn (a
// quarter-turn argument times 64: an exact scaling).  14 fp64 operations + 2 integer operations and no quadrant
// selects, against 17 + ~9 for sincos_quarter_turns<7>.  Cut into stages like the polynomial pair above:
// reduce (issues the table read) -> sin -> cos -> finish.
constexpr int PHASOR_TABLE = 256;
struct TablePhasorStage {
    double z, z2, sn, cs;
    double2 tk;
};
// fill `table` (PHASOR_TABLE entries) with a block of `nthreads` lanes; the caller synchronises
__device__ __forceinline__ void table_phasor_init(double2 *table, int tid, int nthreads)
{
    for (int i = tid; i < PHASOR_TABLE; i += nthreads) {
        double c, sn;
        sincos_quarter_turns<7>((double)i * (4.0 / PHASOR_TABLE), c, sn);
        table[i] = make_double2(c, sn);
    }
}
__device__ __forceinline__ void table_phasor_reduce(TablePhasorStage &s, const double2 *table, double x256)
{
    const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
    const double a = __dadd_rn(x256, MAGIC);
    s.tk = table[__double2loint(a) & (PHASOR_TABLE - 1)];
    s.z = __dsub_rn(x256, __dsub_rn(a, MAGIC));  // [-0.5, 0.5]
    s.z2 = __dmul_rn(s.z, s.z);
}
__device__ __forceinline__ void table_phasor_sin(TablePhasorStage &s)
{
    constexpr double T = 6.283185307179586476925 / PHASOR_TABLE;
    constexpr double S1 = T, S3 = -T * T * T / 6.0, S5 = T * T * T * T * T / 120.0;
    s.sn = __dmul_rn(s.z, fma(s.z2, fma(s.z2, S5, S3), S1));
}
__device__ __forceinline__ void table_phasor_cos(TablePhasorStage &s)
{
    constexpr double T = 6.283185307179586476925 / PHASOR_TABLE;
    constexpr double C2 = -T * T / 2.0, C4 = T * T * T * T / 24.0, C6 = -T * T * T * T * T * T / 720.0;
    s.cs = fma(s.z2, fma(s.z2, fma(s.z2, C6, C4), C2), 1.0);
}
__device__ __forceinline__ void table_phasor_finish(const TablePhasorStage &s, double &c_out, double &s_out)
{
    c_out = fma(s.tk.x, s.cs, -__dmul_rn(s.tk.y, s.sn));
    s_out = fma(s.tk.y, s.cs, __dmul_rn(s.tk.x, s.sn));
}
// the four stages in one piece
__device__ __forceinline__ void table_phasor(const double2 *table, double x256, double &c_out, double &s_out)
{
    TablePhasorStage st;
    table_phasor_reduce(st, table, x256);
    table_phasor_sin(st);
    table_phasor_cos(st);
    table_phasor_finish(st, c_out, s_out);
}


// (cos, sin)(p) for p in RADIANS, for the kernels that must keep the reference's phase p bit for bit
// (phase_delay).  Cody-Waite reduction p = k*(pi/2) + r with pi/2 split in three parts (33 + 33 + 53
// bits, fdlibm's pio2_1/pio2_2/pio2_3): k*part is exact for |k| < 2^20 and each step is one FMA, so
// r carries ~1e-30 of reduction error; f = r*(2/pi) in [-0.5, 0.5] then feeds the 7-term
// quarter-turn polynomials (|err| ~1e-16).  Beyond |p| >= 2^19*pi/2, or for non-finite p, fall back
// to the full-range library sincos.  ~35 fp64 operations against ~90 for the library routine.
__device__ __forceinline__ void sincos_radians(double p, double &c_out, double &s_out)
{
    constexpr double S7[7] = {0x1.921fb54442d18p+0, -0x1.4abbce625be41p-1, 0x1.466bc677587f8p-4,
                              -0x1.32d2cce2e5b19p-8, 0x1.50782fda12d96p-13, -0x1.e30071afc3e59p-19,
                              0x1.e3f38399551bfp-25};
    constexpr double C7[7] = {0x1.0000000000000p+0, -0x1.3bd3cc9be458bp+0, 0x1.03c1f081b0780p-2,
                              -0x1.55d3c7dbfd139p-6, 0x1.e1f4fb60281f6p-11, -0x1.a6c9c1be9eb49p-16,
                              0x1.f3dbcea61b1a4p-22};
    if (!(fabs(p) < 823549.6)) {  // 2^19 * pi/2; also catches NaN / Inf
        sincos(p, &s_out, &c_out);
        return;
    }
    const double MAGIC = 6755399441055744.0;                    // 1.5 * 2^52
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;           // first 33 bits of pi/2
    const double PIO2_2 = 6.07710050630396597660e-11;           // next 33 bits
    const double PIO2_3 = 2.02226624871116645580e-21;           // the rest
    double a = fma(p, TWO_OVER_PI, MAGIC);
    const int q = __double2loint(a);
    const double k = __dsub_rn(a, MAGIC);
    double r = fma(-k, PIO2_1, p);
    r = fma(-k, PIO2_2, r);
    r = fma(-k, PIO2_3, r);
    const double f = __dmul_rn(r, TWO_OVER_PI);
    const double z = __dmul_rn(f, f);
    double ps = S7[6], pc = C7[6];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        ps = fma(ps, z, S7[i]);
        pc = fma(pc, z, C7[i]);
    }
    ps = __dmul_rn(ps, f);
    const bool swap = q & 1;
    double cc = swap ? ps : pc;
    double ss = swap ? pc : ps;
    int chi = __double2hiint(cc) ^ (((q + 1) & 2) << 30);
    int shi = __double2hiint(ss) ^ ((q & 2) << 30);
    c_out = __hiloint2double(chi, __double2loint(cc));
    s_out = __hiloint2double(shi, __double2loint(ss));
}

// exp(-t) for t >= 0 (the Gaussian envelope): t log2(e) = n + f, |f| <= 1/2, 2^-n by v_ldexp_f64, 2^-f = exp(-f ln 2) by
// its Taylor series to degree 11 on |f ln 2| <= 0.347 (next term 2e-14 relative): 17 fp64 operations and four live
// registers, where the library routine's inlined body (~30 operations, a dozen temporaries) put the Gaussian variants
// of the 12-wave kernel over their 168 registers.  The fused chain is checked to 1e-9 against the oracle, not bit for bit.
__device__ __forceinline__ double exp_neg(double t)
{
    const double MAGIC = 6755399441055744.0;   // 1.5 * 2^52
    const double x = __dmul_rn(t, 1.4426950408889634);
    const double a = __dadd_rn(x, MAGIC);
    const int n = __double2loint(a);
    const double f = __dsub_rn(x, __dsub_rn(a, MAGIC));          // [-0.5, 0.5]
    const double y = __dmul_rn(f, -0.6931471805599453);            // exp(y), |y| <= 0.347
    double p = 1.0 / 39916800.0;
    p = fma(p, y, 1.0 / 3628800.0);
    p = fma(p, y, 1.0 / 362880.0);
    p = fma(p, y, 1.0 / 40320.0);
    p = fma(p, y, 1.0 / 5040.0);
    p = fma(p, y, 1.0 / 720.0);
    p = fma(p, y, 1.0 / 120.0);
    p = fma(p, y, 1.0 / 24.0);
    p = fma(p, y, 1.0 / 6.0);
    p = fma(p, y, 0.5);
    p = fma(p, y, 1.0);
    p = fma(p, y, 1.0);
    return t > 1400.0 ? 0.0 : ldexp(p, -n);                        // below the denormals: exactly 0 as exp() gives
}
