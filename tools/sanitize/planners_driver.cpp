// host planners of the fused predict under ASan/UBSan: the bench's arrays, 200 times, identical bytes every time
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>
extern "C" int af_fused_plan_antennas(const int64_t*, const int32_t*, const int32_t*, const double*, int64_t, int64_t, double, int64_t, double*, int32_t*, double*, int*);
extern "C" int af_fused_plan_groups(const int64_t*, const int32_t*, const int32_t*, int64_t, int64_t, int32_t*, int64_t, int64_t*, int32_t*, int64_t, int64_t*);
int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    for (int nant : {64, 128, 7, 61}) {
        const int64_t nrow = nant == 64 ? 1000000 : 150000;
        const int64_t nbl = (int64_t)nant * (nant - 1) / 2, ntime = (nrow + nbl - 1) / nbl;
        std::vector<int64_t> ti(nrow); std::vector<int32_t> a1(nrow), a2(nrow); std::vector<double> uvw(3 * nrow);
        std::mt19937_64 rng(1234 + nant); std::uniform_real_distribution<double> U(-1, 1);
        std::vector<double> xyz(ntime * nant * 3);
        for (auto &x : xyz) x = U(rng) * 2000.0;
        int64_t r = 0;
        for (int64_t t = 0; t < ntime && r < nrow; ++t)
            for (int p = 0; p < nant && r < nrow; ++p)
                for (int q = p + 1; q < nant && r < nrow; ++q, ++r) {
                    ti[r] = t + 17; a1[r] = p; a2[r] = q;
                    for (int c = 0; c < 3; ++c) uvw[3 * r + c] = xyz[(t * nant + p) * 3 + c] - xyz[(t * nant + q) * 3 + c];
                }
        const int64_t nap = 8 * ((nant + 7) / 8);
        std::vector<double> au0, au(ntime * nant * 3); std::vector<int32_t> rm0, rm(ntime * nap * nap);
        std::vector<int32_t> it0, gr0;
        int local_reps = nant == 64 ? reps : reps / 4 + 1;
        for (int k = 0; k < local_reps; ++k) {
            double res; int ok;
            int rc = af_fused_plan_antennas(ti.data(), a1.data(), a2.data(), uvw.data(), nrow, nant, 1e-10, ntime, au.data(), rm.data(), &res, &ok);
            if (rc || !ok) { printf("nant %d: rc %d ok %d res %g\n", nant, rc, ok, res); return 1; }
            int64_t ni = 0, ng = 0;
            rc = af_fused_plan_groups(ti.data(), a1.data(), a2.data(), nrow, nant, nullptr, 0, &ni, nullptr, 0, &ng);
            std::vector<int32_t> items(4 * (ni ? ni : 1)), groups(8 * (ng ? ng : 1));
            rc |= af_fused_plan_groups(ti.data(), a1.data(), a2.data(), nrow, nant, items.data(), ni, &ni, groups.data(), ng, &ng);
            if (rc) { printf("plan_groups rc %d\n", rc); return 1; }
            if (k == 0) { au0 = au; rm0 = rm; it0 = items; gr0 = groups; }
            else if (memcmp(au0.data(), au.data(), au.size() * 8) || memcmp(rm0.data(), rm.data(), rm.size() * 4) || it0 != items || gr0 != groups) {
                printf("nant %d rep %d: plan bytes differ\n", nant, k); return 1;
            }
        }
        printf("nant %d: %d identical plans (residual ok)\n", nant, local_reps);
    }
    return 0;
}
