/*
 * oracle/knn_oracle.c — CPU restatement of simple_knn._C.distCUDA2 (reference call sites:
 * /root/reference/scene/gaussian_model.py:213,641).
 *
 * TEST INFRASTRUCTURE ONLY (same rule as raster_oracle.c: never imported by the product path).
 *
 * PARITY UNPINNED: the implementation lives in the un-vendored submodule submodules/simple-knn
 * (/root/reference/.gitmodules:1-3, directory empty).  Restated from the published algorithm
 * (upstream:simple-knn/simple_knn.cu): for every point, the exact three nearest
 * OTHER indices (duplicates count with distance 0), result = (d1^2 + d2^2 + d3^2) / 3 with the three
 * squared distances held sorted ascending and summed in that order.  Upstream prunes with Morton-ordered
 * boxes; the k-best set is unique as a multiset, so an exhaustive scan gives bit-identical fp32 output as
 * long as each squared distance is evaluated as dx*dx + dy*dy + dz*dz without FMA contraction.
 * Independent check in tests: scipy.spatial.cKDTree.query(k=4) in float64.
 */
#include <float.h>
#include <stdint.h>

static inline void update_kbest3(float d, float *knn) {
    /* upstream:simple_knn.cu updateKBest<3> */
    for (int j = 0; j < 3; j++) {
        if (knn[j] > d) {
            float t = knn[j];
            knn[j] = d;
            d = t;
        }
    }
}

void oracle_knn3_mean_dist2(int P, const float *pts, float *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            float dx = px - pts[3 * j], dy = py - pts[3 * j + 1], dz = pz - pts[3 * j + 2];
            float d = dx * dx + dy * dy + dz * dz;
            update_kbest3(d, best);
        }
        out[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}
