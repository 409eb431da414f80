// Host check of the 32-byte node encoder the kernels' BVHs are made with (csrc/rtc_qnode.hpp): whatever two child
// boxes go in -- tiny, huge, degenerate, far from the origin, one child absent -- the boxes a node decodes to must
// CONTAIN the boxes it was made from (the hierarchy stays conservative) and be no looser than two quantisation steps
// of the node's frame plus the grid step of the tree.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../../ray_tracer_challenge_amd/csrc/rtc_qnode.hpp"

int main() {
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const int32_t EMPTY = INT32_MIN;
    long checked = 0;
    for (int tree = 0; tree < 400; tree++) {
        // a tree box of random size and position, from 1e-3 to 1e6 across, up to 1e6 from the origin
        double size = std::pow(10.0, -3.0 + 9.0 * U(rng)), centre = (U(rng) - 0.5) * 2.0 * std::pow(10.0, 6.0 * U(rng));
        float tlo[3], thi[3];
        for (int k = 0; k < 3; k++) { tlo[k] = (float)(centre - size * (0.2 + U(rng))); thi[k] = (float)(centre + size * (0.2 + U(rng))); }
        rtc::TreeGrid G = rtc::make_tree_grid(tlo, thi);
        for (int k = 0; k < 3; k++) if (!(G.lo[k] <= tlo[k]) || !(G.cell[k] > 0)) { printf("FAIL grid\n"); return 1; }
        for (int node = 0; node < 500; node++) {
            float box[12];
            double frac = std::pow(10.0, -6.0 * U(rng));       // node extent relative to the tree, down to 1e-6
            for (int c = 0; c < 2; c++)
                for (int k = 0; k < 3; k++) {
                    double ext = (double)thi[k] - (double)tlo[k];
                    double a = (double)tlo[k] + U(rng) * ext * (1.0 - frac), b = a + U(rng) * ext * frac;
                    if (node % 17 == 0) b = a;                  // degenerate (flat) boxes
                    box[6 * c + k] = (float)a; box[6 * c + 3 + k] = (float)b;
                    if (box[6 * c + k] < tlo[k]) box[6 * c + k] = tlo[k];
                    if (box[6 * c + 3 + k] > thi[k]) box[6 * c + 3 + k] = thi[k];
                    if (box[6 * c + 3 + k] < box[6 * c + k]) box[6 * c + 3 + k] = box[6 * c + k];
                }
            int32_t c0 = 5, c1 = node % 11 == 0 ? EMPTY : ~((7 << 4) | 1);
            rtc::QNode q;
            rtc::qnode_encode(box, c0, c1, EMPTY, G, q);
            if ((int32_t)q.w[6] != c0 || (int32_t)q.w[7] != c1) { printf("FAIL children\n"); return 1; }
            double out[12];
            rtc::qnode_decode(q, G, out);
            for (int c = 0; c < 2; c++) {
                if (c == 1 && c1 == EMPTY) {
                    // lo = 255, hi = 0 on every axis: bytes 2, 3 of w4 and all of w5
                    if ((q.w[4] >> 16) != 0xFFFFu || q.w[5] != 0x000000FFu) { printf("FAIL empty child not inverted\n"); return 1; }
                    continue;
                }
                for (int k = 0; k < 3; k++) {
                    double frame = 0;                          // the node's own extent on this axis
                    double L = box[k], H = box[3 + k];
                    if (c1 != EMPTY) { L = std::fmin(L, box[6 + k]); H = std::fmax(H, box[9 + k]); }
                    frame = H - L;
                    double step = 2.0 * (frame + 2.0 * G.cell[k]) / 255.0 * 2.0 + 2.0 * G.cell[k] + 1e-30;
                    double lo = out[6 * c + k], hi = out[6 * c + 3 + k];
                    if (lo > box[6 * c + k] || hi < box[6 * c + 3 + k]) { printf("FAIL containment tree %d node %d\n", tree, node); return 1; }
                    if (box[6 * c + k] - lo > step * 1.01 + 1e-6 * std::fabs(lo) || hi - box[6 * c + 3 + k] > step * 1.01 + 1e-6 * std::fabs(hi)) {
                        printf("FAIL loose tree %d node %d axis %d: in [%g, %g] out [%g, %g] frame %g cell %g\n", tree, node, k, box[6 * c + k], box[6 * c + 3 + k], lo, hi, frame, G.cell[k]);
                        return 1;
                    }
                    checked++;
                }
            }
        }
    }
    printf("OK %ld planes\n", checked);
    return 0;
}
