// dev tool (host only, runs without a GPU): the Gram work partition of a shape -- types, staged block rows, slices, the row traffic
// the plan implies and the busiest SIMD against the mean.   usage: plan_dump p n J f64(0|1) [workgroup budget of the second launch]
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/plan_dump tools/plan_dump.hip
#include "../ces_amd/csrc/kernels_gram.hip"
#include <cstdio>
using namespace cesx;
int main(int argc, char** argv) {
    int p = atoi(argv[1]), n = atoi(argv[2]); long long J = atoll(argv[3]); int f64 = atoi(argv[4]); int budget_b = argc > 5 ? atoi(argv[5]) : 248;
    int tile = f64 ? 16 : 32, nbw = f64 ? 8 : 4, kt = f64 ? 16 : 32, esz = f64 ? 8 : 4;
    int P = p + n, pbU = (p + tile - 1) / tile;
    for (int part = 0; part < 2; ++part) {
        GramPlan pl = make_gram_plan(P, tile, nbw, MAX_STAGE_ROWS, part + 1, pbU, 1, part == 0 ? 256 : budget_b, J / kt);
        double traffic = 0, slabs = 0; long long worst = 0, sum = 0;
        printf("part %d: %d types, %d wgs, %d blocks, max_rb %d\n", part, pl.ntypes, pl.total_wgs, pl.nblocks, pl.max_rb);
        for (int t = 0; t < pl.ntypes; ++t) {
            const int* h = &pl.type_hdr[t * 8];
            printf("  type %2d: nrb %2d blocks %3d slices %3d  (blocks/rowblock %.2f)\n", t, h[0], h[3], h[5], (double)h[3] / h[0]);
            traffic += (double)h[0] * tile * J * esz;
            { long long nt_ = J / kt, tps = (nt_ + h[5] - 1) / h[5]; int mxs = (h[3] + 15) / 16 * 4; /* approx per-SIMD blocks */
              int per_simd = (h[3] + 3) / 4; worst = std::max(worst, tps * per_simd); sum += (long long)h[3] * nt_; (void)mxs; }
            slabs += (double)h[5] * h[3] * tile * tile * esz;
        }
        printf("  busiest SIMD %lld block-tiles, mean %.0f (per SIMD over %d WGs) -> efficiency %.3f\n", worst, (double)sum / 4 / pl.total_wgs, pl.total_wgs, (double)sum / 4 / pl.total_wgs / worst);
        printf("  row traffic %.1f MB (algorithmic %.1f MB), slabs %.1f MB\n", traffic / 1e6, (double)P * J * esz / 1e6, slabs / 1e6);
    }
}
namespace cesx { int launch_gram2(Engine&, int, const void*, const void*, hipStream_t) { return -1; } }
