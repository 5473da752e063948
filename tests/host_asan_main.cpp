// Host-side AddressSanitizer run of the C-ABI (aux: "an -fsanitize=address host build"): the library's host code
// (argument validation, layout arithmetic, error strings, the profiler's bookkeeping) instrumented with ASan — device code
// untouched (-fno-gpu-sanitize) — and driven through every entry point that does not need a GPU.  Built and run by
// tests/test_host_asan.py; exits non-zero on a wrong answer, ASan aborts on a bad access.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "../include/das3r_raster.h"

static char *no_alloc(void *, size_t) { return nullptr; }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "host_asan: %s failed (line %d): %s\n", #c, __LINE__, das3r_last_error()); return 1; } } while (0)

int main() {
    CHECK(das3r_abi_version() == DAS3R_ABI_VERSION);
    // layout arithmetic over a sweep of shapes: every offset inside its buffer, 256-byte aligned, monotone in the capacity
    for (int P : {0, 1, 255, 256, 257, 100000, 5000000}) {
        for (long long I : {0ll, 1ll, 262571ll, 2614506ll, 40000000ll}) {
            for (int W : {1, 48, 123, 512, 1920, 4097}) {
                const int H = W == 1920 ? 1080 : (W * 9 + 15) / 16;
                das3r_raster_layout L;
                memset(&L, 0xff, sizeof(L));
                CHECK(das3r_raster_get_layout(P, I, W, H, &L) == DAS3R_OK);
                CHECK(L.xy < L.geom_bytes && L.conic_opacity == L.xy + 16 && L.rgbd == L.xy + 32 && L.splat_stride == 64);
                CHECK(L.point_list < L.binning_bytes && L.final_T < L.img_bytes && L.n_contrib < L.img_bytes && L.ranges < L.img_bytes);
                CHECK(L.xy % 256 == 0 && L.point_list % 256 == 0 && L.ranges % 256 == 0);
                das3r_raster_layout L2;
                CHECK(das3r_raster_get_layout(P, I + 1000, W, H, &L2) == DAS3R_OK && L2.binning_bytes >= L.binning_bytes && L2.geom_bytes == L.geom_bytes);
            }
        }
    }
    das3r_raster_layout L;
    CHECK(das3r_raster_get_layout(-1, 0, 16, 16, &L) == DAS3R_ERR_INVALID_ARG && strlen(das3r_last_error()) > 0);
    CHECK(das3r_raster_get_layout(1, 0, 0, 16, &L) == DAS3R_ERR_INVALID_ARG);
    CHECK(das3r_raster_get_layout(1, 0, 16, 16, nullptr) == DAS3R_ERR_INVALID_ARG);
    CHECK(das3r_raster_backward_scratch_bytes(0) == 36 + 16 && das3r_raster_backward_scratch_bytes(1000) == 36000 + 16);
    // argument validation of the forward / backward entry points (everything is rejected before a device is touched)
    das3r_raster_args a;
    memset(&a, 0, sizeof(a));
    das3r_raster_in in;
    memset(&in, 0, sizeof(in));
    das3r_raster_out out;
    memset(&out, 0, sizeof(out));
    das3r_raster_saved saved;
    memset(&saved, 0, sizeof(saved));
    CHECK(das3r_raster_forward(nullptr, &in, &out, no_alloc, no_alloc, no_alloc, nullptr, &saved, nullptr) == DAS3R_ERR_INVALID_ARG);
    a.P = 10; a.image_width = 0; a.image_height = 16;
    CHECK(das3r_raster_forward(&a, &in, &out, no_alloc, no_alloc, no_alloc, nullptr, &saved, nullptr) == DAS3R_ERR_INVALID_ARG);
    a.image_width = 16; a.tanfovx = a.tanfovy = 0.5f; a.M = 16; a.sh_degree = 3;
    float dummy[64] = {0};
    a.bg = a.viewmatrix = a.projmatrix = a.campos = dummy;
    in.means3D = in.opacities = dummy;
    CHECK(das3r_raster_forward(&a, &in, &out, no_alloc, no_alloc, no_alloc, nullptr, &saved, nullptr) == DAS3R_ERR_INVALID_ARG);   // neither SHs nor colours
    CHECK(strstr(das3r_last_error(), "excatly one of either SHs or precomputed colors") != nullptr);
    in.shs = dummy; in.colors_precomp = dummy;
    CHECK(das3r_raster_forward(&a, &in, &out, no_alloc, no_alloc, no_alloc, nullptr, &saved, nullptr) == DAS3R_ERR_INVALID_ARG);   // both
    in.colors_precomp = nullptr; in.scales = dummy;
    CHECK(das3r_raster_forward(&a, &in, &out, no_alloc, no_alloc, no_alloc, nullptr, &saved, nullptr) == DAS3R_ERR_INVALID_ARG);   // scales without rotations
    CHECK(strstr(das3r_last_error(), "exactly one of either scale/rotation pair or precomputed 3D covariance") != nullptr);
    in.rotations = dummy; a.sh_degree = 4;
    CHECK(das3r_raster_forward(&a, &in, &out, no_alloc, no_alloc, no_alloc, nullptr, &saved, nullptr) == DAS3R_ERR_INVALID_ARG);   // degree out of range
    a.sh_degree = 3; a.M = 9;
    CHECK(das3r_raster_forward(&a, &in, &out, no_alloc, no_alloc, no_alloc, nullptr, &saved, nullptr) == DAS3R_ERR_INVALID_ARG);   // M too small for the degree
    a.M = 16;
    CHECK(das3r_raster_forward(&a, &in, nullptr, no_alloc, no_alloc, no_alloc, nullptr, &saved, nullptr) == DAS3R_ERR_INVALID_ARG);   // no outputs
    das3r_raster_grads g;
    memset(&g, 0, sizeof(g));
    CHECK(das3r_raster_backward(&a, &in, nullptr, dummy, &g, nullptr) == DAS3R_ERR_INVALID_ARG);
    CHECK(das3r_raster_check(nullptr, nullptr) == DAS3R_ERR_INVALID_ARG);
    CHECK(das3r_raster_check(&saved, nullptr) == DAS3R_OK);   // no ticket: nothing to examine
    CHECK(das3r_mark_visible(-1, nullptr, nullptr, nullptr, nullptr, nullptr) == DAS3R_ERR_INVALID_ARG);
    // profiler bookkeeping and counters without a device
    das3r_profile_enable(1);
    das3r_profile_enable(0);
    char buf[64];
    CHECK(das3r_profile_report(buf, sizeof(buf)) == 0 && buf[0] == 0);
    uint64_t st[4] = {9, 9, 9, 9};
    das3r_get_stats(st);
    CHECK(st[3] == 0);
    das3r_reload_switches();
    CHECK(das3r_has_experiments() == 0 || das3r_has_experiments() == 1);
    printf("host_asan: all checks passed\n");
    return 0;
}
