// TEST INFRASTRUCTURE: host build of csrc/mvmc_trf_faithful.h for checks against SciPy (tests/test_trf_faithful_cpu.py)
#include <vector>
#include <string.h>
#include "../../multiview_motion_capture_amd/csrc/mvmc_trf_faithful.h"
using namespace trf_faithful;
extern "C" int trf_check_ik(const double* dirs, const int* parents, const int* side_map, int n_side, const double* pose18, const double* Pm,
                            int nv, const double* side_fixed, int stage, int max_nfev, double* x, double* out4) {
    Skel sk;
    memcpy(sk.dirs, dirs, sizeof(sk.dirs));
    for (int j = 0; j < 18; ++j) { sk.parents[j] = parents[j]; sk.side_map[j] = side_map[j]; }
    sk.n_side = n_side;
    IkResidual fun{&sk, pose18, Pm, side_fixed, nv, stage};
    const int n = stage == 0 ? 57 : 57 + n_side, m = fun.m();
    std::vector<double> work(work_doubles(m, n));
    Result r = trf(Serial(), fun, n, m, x, max_nfev, work.data());
    out4[0] = r.cost; out4[1] = r.nfev; out4[2] = r.status; out4[3] = r.njev;
    return 0;
}
extern "C" int trf_check_postopt(const double* pose, const double* Pm, int nv, int n_pts, int max_nfev, double* x, double* out4) {
    PostoptResidual fun{pose, Pm, nv, n_pts};
    const int n = 3 * n_pts, m = fun.m();
    std::vector<double> work(work_doubles(m, n));
    Result r = trf(Serial(), fun, n, m, x, max_nfev, work.data());
    out4[0] = r.cost; out4[1] = r.nfev; out4[2] = r.status; out4[3] = r.njev;
    return 0;
}
extern "C" int trf_check_svd(int m, int n, double* A /* col-major, ld = max(m,n) */, double* f, double* s, double* suf, double* V) {
    const int mm = m > n ? m : n;
    std::vector<double> fq(f, f + mm);
    svd_pieces(Serial(), m, n, A, mm, fq.data(), s, suf, V);
    return 0;
}
extern "C" int trf_check_triangulate18(const double* pose18, const double* Pm, int nv, double* x54, double* dlt54) {
    for (int j = 0; j < 18; ++j) dlt_point(pose18, Pm, nv, j, 0.01, dlt54 + 3 * j);
    std::vector<double> work(work_doubles(18 * nv, 54));
    triangulate_postopt18(Serial(), pose18, Pm, nv, x54, work.data());
    return 0;
}
