// The planner of the short-series launch (small_plan, csrc/ngp_internal.h) over every geometry it can
// be asked about: whatever it accepts must fit the kernel's fixed resources — twenty register blocks
// per wave (seven workers in a main sweep, eight waves otherwise), SM_MAX_PANEL panel blocks in LDS,
// at most SM_MAX_SWEEPS sweeps — and must carry every aux row-block exactly once; the inverse phase of
// a gradient job must give every block column to one wave.  Host code only (hipcc --cuda-host-only).
#include <cstdio>
#include <vector>

#include "../../nowcastautogp_amd/csrc/ngp_internal.h"

using namespace ngp;

static int failures = 0;
#define CHECK(c, ...) do { if (!(c)) { ++failures; printf("FAIL %s: ", #c); printf(__VA_ARGS__); printf("\n"); } } while (0)

static int col_count(const SmallSweep &sw, int nbe, int k) {
    const int cm = sw.main == 1 ? nbe - 1 - k : 0;
    const int ci = std::max(std::min(sw.i1, k + 1) - sw.i0, 0);
    return cm + ci + (sw.a1 - sw.a0);
}

int main() {
    int accepted = 0, refused = 0;
    for (int grad = 0; grad < 2; ++grad)
        for (int n0 = 64; n0 <= 320; n0 += 64)
            for (int n_real = grad ? n0 - 63 : n0; n_real <= n0; ++n_real)
                for (int naux = 1; naux <= (grad ? 1 : NGP_MAX_AUX); ++naux) {
                    JobGeom g{};
                    g.n0 = n0;
                    g.nb0 = n0 / NB;
                    g.n_real = n_real;
                    g.aux_identity = grad;
                    g.naux = grad ? n0 + 1 : naux;
                    g.naux_pad = grad ? n0 + NB : (naux + NB - 1) / NB * NB;
                    g.short_series = 1;
                    SmallPlan pl{};
                    if (!small_plan(g, &pl)) {
                        ++refused;
                        CHECK(n0 > 256 || !grad, "a gradient geometry of n0 = %d, n_real = %d was refused", n0, n_real);
                        continue;
                    }
                    ++accepted;
                    CHECK(n0 <= 256, "n0 = %d accepted", n0);
                    const int nbe = pl.nbe, nb16 = n0 / 16;
                    CHECK(nbe == (n_real + 15) / 16 && nbe <= 16, "nbe %d", nbe);
                    CHECK(pl.nsweeps >= 1 && pl.nsweeps <= SM_MAX_SWEEPS, "sweeps %d", pl.nsweeps);
                    CHECK(pl.npanel <= SM_MAX_PANEL && small_lds_bytes(pl) <= 160 * 1024, "panel %d", pl.npanel);
                    CHECK(pl.sw[0].main == 1, "the first sweep factorises");
                    std::vector<int> dense(nb16 + 64, 0);
                    int n_inverse = 0;
                    for (int si = 0; si < pl.nsweeps; ++si) {
                        const SmallSweep &sw = pl.sw[si];
                        CHECK(si == 0 || sw.main != 1, "one main sweep");
                        if (sw.main == 2) {
                            ++n_inverse;
                            CHECK(grad && sw.i0 == 0 && sw.i1 == nbe, "inverse phase covers the identity rows");
                            for (int j = 0; j < nbe; ++j) {
                                const int w = (int)((pl.colwave >> (4 * j)) & 15);
                                CHECK(w >= 0 && w < SM_WAVES, "column %d on wave %d", j, w);
                            }
                            continue;
                        }
                        int blocks = 0;
                        for (int k = 0; k < nbe; ++k) blocks += col_count(sw, nbe, k);
                        const int cap = (sw.main == 1 ? SM_WAVES - 1 : SM_WAVES) * SM_NSLOT;
                        CHECK(blocks <= cap, "n0 %d n_real %d naux %d sweep %d: %d blocks > %d", n0, n_real, g.naux, si, blocks, cap);
                        CHECK(nbe + (sw.i1 - sw.i0) + (sw.a1 - sw.a0) <= pl.npanel, "panel rows of sweep %d", si);
                        for (int a = sw.a0; a < sw.a1; ++a) ++dense[(size_t)a];
                    }
                    if (grad) {
                        CHECK(n_inverse == 1, "one inverse phase");
                        CHECK(dense[(size_t)nb16] == 1, "y' row-block carried %d times", dense[(size_t)nb16]);
                    } else {
                        const int nba = (naux + 15) / 16;
                        for (int a = 0; a < nba; ++a) CHECK(dense[(size_t)a] == 1, "aux row-block %d carried %d times", a, dense[(size_t)a]);
                        for (int a = nba; a < (int)dense.size(); ++a) CHECK(dense[(size_t)a] == 0, "aux row-block %d is not there", a);
                    }
                }
    printf("%d geometries accepted, %d refused, %d failures\n", accepted, refused, failures);
    return failures ? 1 : 0;
}
