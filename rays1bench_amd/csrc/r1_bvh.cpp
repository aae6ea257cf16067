// r1_bvh.cpp — host-side builder of the optional spatial index.
//
// The reference has no acceleration structure ("it's even more insane idea to use linear
// searching", README.md:163; the nested-Hitable stub at rayweek1.cpp:328-349 is unused): every
// ray is tested against every sphere.  R1_VARIANT_BVH keeps the reference's per-sphere
// arithmetic (exact_offer in r1_trace.hpp = rayweek1.cpp:192-202, :294-313) and only decides
// WHICH spheres are presented to it, through a binary tree of axis-aligned boxes that is
// conservative with respect to the reference's fp32 test, so that pixels and ray counts stay
// bit-identical to the exhaustive sweep (tests/test_gpu_parity.py::test_bvh_*).
//
// Conservativeness.  Let u = 2^-24, v = c - o.  The reference's fp32 discriminant differs from
// the real-number one by E1 <= 23 u |v|^2 + 2 u r^2 (co rounding, two FMA chains, the square,
// the final subtraction, and |d| = 1 +- 3u).  A sphere whose reference discriminant has a clear
// sign bit therefore has its centre within sqrt(r^2 + E1) <= r_eff + E1 / (2 r_eff) of the
// ray's line (r_eff = max(r, r_floor); the r_floor/2 this costs a degenerate sphere is added
// to the pad), and the point at its offered t lies within that distance + 5 u |v| of the
// centre.  A child box is stored as centre m and half extent e and has to be inflated by at least
//      pad_min = w2 * |m - o|^2 + k
// with  w2 = 2 * (40 u * max_i 1 / (2 r_eff,i)) + 2^-19   and   k = w2 * h^2 + 2^-20 + ...,
// h = max_i |c_i - m|, using |v|^2 <= 2 |m - o|^2 + 2 h^2.  The 2^-19 / 2^-20 terms cover the
// rounding of the slab test itself: products and differences (<= 6u D in position units, D the
// largest distance involved) and the 1-ulp reciprocals of the direction (v_rcp_f32, <= 2^-22 D),
// together < 2^-20.7 D <= 2^-21.7 (1 + D^2), against a budget of 2^-20 (1 + D^2).
//
// What the kernel evaluates (r1_trace.hpp::bvh_box) is ONE fused multiply-add per node,
//      pad = A * R2 + K,      R2 = |o - C|^2 computed once per ray,
// C a fixed point of the scene (the per-axis median of the sphere centres, R1Bvh::centre), and
// A = 2 w2 + u, K = k + 2 w2 g^2 + u (1 + |C|_1), g = the larger |m - C| of the node's two children:
// |m - o|^2 <= 2 |o - C|^2 + 2 |m - C|^2, so pad >= pad_min for both children, at the price of boxes
// a few 1e-2 units wider than necessary (large scene: pad ~0.03 next to r = 0.45).  The u terms pay for
// the slab test's own form a = m * (1/d) - o * (1/d) (fused, o * (1/d) rounded once per ray): next to
// the errors budgeted above it adds u |o| (1 + u) in position units, and |o| <= |o - C| + |C| <=
// (1 + R2) / 2 + |C|.  A and K are rounded up with 2^-18 relative headroom for the fp32 evaluation
// of R2 (3 u) and of the fused multiply-add (u).
//
// Scenes of small spheres (R1Bvh::pad_local; the 100 004-sphere lattice: r = 0.034, spacing 0.08): w2 grows with 1 / r and
// a pad measured from ONE point of the scene (~0.1 there) would bury the boxes.  Such trees measure the distance per node:
//      pad = A * |s|^2 + K,   s = m0 + m1 - 2 o = 2 (m_n - o),  m_n the midpoint of the two child centres,
// A = (1 + eps) w2 / 4 + u / 8, K = k + (1 + 1 / eps) w2 g^2 + u (1/2 + |m_n|) + ..., g = |m0 - m1| / 2 = |m_c - m_n|:
// |m_c - o|^2 <= (|m_n - o| + g)^2 <= (1 + eps) |s|^2 / 4 + (1 + 1 / eps) g^2 for every eps > 0 (eps = g / the scene's
// scale: tight for origins that far away), and |o| <= |s| / 2 + |m_n| <= (1 + |s|^2 / 4) / 2 + |m_n|.  s is computed in
// fp32 (error per component <= delta = 4 u (|m0|_inf + |m1|_inf + 2 |o|_inf) <= 4 u (3 M + |s|), M = |m0|_inf + |m1|_inf);
// |s|^2 <= (1 + 2^-10) |s'|^2 + (1 + 2^10) 3 delta^2 puts 2^-10 more on A and A 3075 * 288 u^2 M^2 on K (the |s|^2
// part of delta^2 is 1e-10 relative).  9 VALU instructions per node visit instead of 1, still 7 fewer than round 1's form.
// The choice is made per tree from the median radius and the median distance from C: local when 4 w2_med G^2 > r_med / 4.
// Every constant is rounded up.  Extra visits are harmless: leaves apply the reference's rule.
#include <hip/hip_runtime.h>

#include "r1_device.h"
#include "../../include/rays1.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

struct R1Bvh
{
    std::vector<float> nodes;   // 16 floats per node, see R1DeviceScene::bvh_nodes
    std::vector<float> prims;   // leaf order, 8 floats per PAIR of spheres: {cx_a cx_b cy_a cy_b cz_a cz_b rsq_a rsq_b}
    std::vector<uint32_t> ids;  // 2 per pair: active index, 0xFFFFFFFF for the partner of an odd sphere
    int max_depth = 0;          // inner nodes on the longest root-to-leaf path
    uint32_t n_leaves = 0;
    float centre[3] = {0, 0, 0}; // C of the pad formula (see above)
    int pad_local = 0;           // 1: pad measured per node (scenes of small spheres), 0: from `centre`
    int root_leaf = 0;           // 1 / 2: child 0 / 1 of the root is a leaf of <= 2 sphere pairs and the other child an inner node (the root step), 0: no
};

namespace
{

struct Box
{
    double lo[3], hi[3];   // of the spheres' extents c +- r
    double clo[3], chi[3]; // of the centres
    double kmax;           // max 1 / (2 r_eff)
    double rmax;           // max r
    double floor_pad;      // max r_floor / 2 over degenerate members
    void clear()
    {
        for (int a = 0; a < 3; ++a)
            lo[a] = clo[a] = 1e300, hi[a] = chi[a] = -1e300;
        kmax = rmax = floor_pad = 0;
    }
    void merge(const Box &b)
    {
        for (int a = 0; a < 3; ++a)
        {
            lo[a] = std::min(lo[a], b.lo[a]), hi[a] = std::max(hi[a], b.hi[a]);
            clo[a] = std::min(clo[a], b.clo[a]), chi[a] = std::max(chi[a], b.chi[a]);
        }
        kmax = std::max(kmax, b.kmax), rmax = std::max(rmax, b.rmax), floor_pad = std::max(floor_pad, b.floor_pad);
    }
    double area() const
    {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

float round_up(double v)
{
    float f = (float)v;
    if ((double)f < v)
        f = nextafterf(f, INFINITY);
    return nextafterf(f, INFINITY);
}

struct Builder
{
    const float *cx, *cy, *cz, *rsq; // active order, the fp32 values the exact test reads
    const double *rbound;             // radius the boxes must cover (r1_bound_radius)
    std::vector<Box> sphere;
    std::vector<uint32_t> order;
    R1Bvh *out;
    int leaf_max;
    double centre[3]; // C of the pad formula, == out->centre
    bool pad_local = false;
    double d_typ = 1.0; // pad_local: the distance at which the per-node pad is tight (twice the median distance of the centres from C)

    static const uint32_t LEAF = 0x80000000u;

    Box bound(uint32_t b, uint32_t e) const
    {
        Box r;
        r.clear();
        for (uint32_t i = b; i < e; ++i)
            r.merge(sphere[order[i]]);
        return r;
    }

    // child box -> {m, e} in fp32 covering [lo, hi], and this child's (w2, k)
    static void encode(const Box &bx, float m[3], float e[3], double &w2, double &k)
    {
        double h2 = 0;
        for (int a = 0; a < 3; ++a)
        {
            m[a] = (float)(0.5 * (bx.lo[a] + bx.hi[a]));
            e[a] = round_up(std::max(bx.hi[a] - (double)m[a], (double)m[a] - bx.lo[a]));
            const double hc = std::max(std::fabs(bx.clo[a] - (double)m[a]), std::fabs(bx.chi[a] - (double)m[a]));
            h2 += hc * hc;
        }
        const double u = ldexp(1.0, -24);
        w2 = 2.0 * (40.0 * u * bx.kmax) + ldexp(1.0, -19);
        k = w2 * h2 + ldexp(1.0, -20) + 4.0 * u * bx.rmax + bx.floor_pad;
    }

    uint32_t make_leaf(uint32_t b, uint32_t e)
    {
        // spheres are stored in PAIRS (two spheres per pair of 16-byte loads); the partner of
        // an odd sphere has radius_sq = -inf, which makes its discriminant -inf: never flagged
        const uint32_t first = (uint32_t)(out->ids.size() / 2);
        const uint32_t pairs = (e - b + 1) / 2;
        for (uint32_t q = 0; q < pairs; ++q)
        {
            const uint32_t ia = order[b + 2 * q];
            const bool has_b = b + 2 * q + 1 < e;
            const uint32_t ib = has_b ? order[b + 2 * q + 1] : ia;
            const float pad_r = -INFINITY;
            const float v[8] = {cx[ia], has_b ? cx[ib] : 0.0f, cy[ia], has_b ? cy[ib] : 0.0f,
                                cz[ia], has_b ? cz[ib] : 0.0f, rsq[ia], has_b ? rsq[ib] : pad_r};
            out->prims.insert(out->prims.end(), v, v + 8);
            out->ids.push_back(ia);
            out->ids.push_back(has_b ? ib : 0xFFFFFFFFu);
        }
        ++out->n_leaves;
        return LEAF | (pairs << 28) | first;
    }

    // returns the child reference for order[b, e); bx = its bounds
    uint32_t build(uint32_t b, uint32_t e, int depth, const Box &bx)
    {
        const uint32_t n = e - b;
        if (n <= (uint32_t)leaf_max)
            return make_leaf(b, e);

        // levels a balanced split of n spheres still needs; SAH only while the depth budget allows
        int need = 0;
        while (((uint64_t)leaf_max << need) < n)
            ++need;
        const bool force_median = depth + need + 1 >= R1_BVH_STACK;

        uint32_t mid = 0;
        int axis = 0;
        // Outliers first: a few spheres much larger than the rest of the node (the ground and the r = 2 balls among the
        // r = 0.45 lattice of the reference's large scene) would inflate every box on their way down a centroid split
        // (the balls sat four levels deep and made the boxes above them 2.7 high instead of 0.5).  Up to `leaf_max`
        // spheres more than R1_BVH_PEEL_RATIO (4) x the node's median radius become ONE leaf here and the rest keeps tight
        // boxes: large scene 24.3 -> 26.5 Grays/s, 100 004-sphere scene 12.2 -> 12.8 (tools: R1_BVH_PEEL=0 switches it off).
        static const int peel_env = (int)r1_knob("R1_BVH_PEEL", 1);
        if (peel_env && !force_median && depth + need + 2 < R1_BVH_STACK && n > 2u * (uint32_t)leaf_max) // (a peel costs a level)
        {
            std::vector<double> rs(n);
            for (uint32_t i = 0; i < n; ++i)
                rs[i] = sphere[order[b + i]].rmax;
            std::nth_element(rs.begin(), rs.begin() + n / 2, rs.end());
            static const double ratio_env = r1_knob_f("R1_BVH_PEEL_RATIO", 4.0);
            const double big = ratio_env * rs[n / 2];
            uint32_t nb = 0;
            for (uint32_t i = b; i < e; ++i)
                nb += sphere[order[i]].rmax > big ? 1u : 0u;
            if (nb >= 1 && nb <= (uint32_t)leaf_max)
            {
                std::stable_partition(order.begin() + b, order.begin() + e, [&](uint32_t q) { return sphere[q].rmax > big; });
                mid = b + nb;
            }
        }
        if (mid == 0)
        {
            double ext[3];
            for (int a = 0; a < 3; ++a)
                ext[a] = bx.chi[a] - bx.clo[a];
            axis = ext[1] > ext[0] ? 1 : 0;
            if (ext[2] > ext[axis])
                axis = 2;
        }
        if (!force_median && mid == 0)
        {
            // binned surface-area heuristic over the three axes (16 bins on the centres)
            const int NB = 16;
            double best = 1e300;
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; ++a)
            {
                const double lo = bx.clo[a], ext = bx.chi[a] - bx.clo[a];
                if (!(ext > 0))
                    continue;
                Box bins[NB];
                uint32_t cnt[NB];
                for (int i = 0; i < NB; ++i)
                    bins[i].clear(), cnt[i] = 0;
                const double scale = NB / ext;
                for (uint32_t i = b; i < e; ++i)
                {
                    const Box &s = sphere[order[i]];
                    int bi = (int)((s.clo[a] - lo) * scale);
                    bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
                    bins[bi].merge(s), ++cnt[bi];
                }
                double right_area[NB];
                uint32_t right_cnt[NB];
                Box acc;
                acc.clear();
                uint32_t c = 0;
                for (int i = NB - 1; i > 0; --i)
                {
                    if (cnt[i])
                        acc.merge(bins[i]);
                    c += cnt[i];
                    right_area[i] = c ? acc.area() : 0, right_cnt[i] = c;
                }
                acc.clear(), c = 0;
                for (int i = 0; i < NB - 1; ++i)
                {
                    if (cnt[i])
                        acc.merge(bins[i]);
                    c += cnt[i];
                    if (c == 0 || right_cnt[i + 1] == 0)
                        continue;
                    const double cost = acc.area() * c + right_area[i + 1] * right_cnt[i + 1];
                    if (cost < best)
                        best = cost, best_axis = a, best_bin = i;
                }
            }
            if (best_axis >= 0)
            {
                const int a = best_axis;
                const double lo = bx.clo[a], scale = NB / (bx.chi[a] - bx.clo[a]);
                auto it = std::partition(order.begin() + b, order.begin() + e, [&](uint32_t s) {
                    int bi = (int)((sphere[s].clo[a] - lo) * scale);
                    bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
                    return bi <= best_bin;
                });
                mid = (uint32_t)(it - order.begin());
            }
        }
        if (mid <= b || mid >= e)
        {
            // median along the widest axis of the centres (also: all centres equal)
            mid = b + n / 2;
            const int a = axis;
            std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](uint32_t p, uint32_t q) {
                return sphere[p].clo[a] < sphere[q].clo[a] || (sphere[p].clo[a] == sphere[q].clo[a] && p < q);
            });
        }

        const uint32_t node = (uint32_t)(out->nodes.size() / 16);
        out->nodes.resize(out->nodes.size() + 16);
        out->max_depth = std::max(out->max_depth, depth + 1);
        const Box b0 = bound(b, mid), b1 = bound(mid, e);
        const uint32_t c0 = build(b, mid, depth + 1, b0);
        const uint32_t c1 = build(mid, e, depth + 1, b1);
        fill(node, b0, c0, b1, c1);
        return node;
    }

    void fill(uint32_t node, const Box &b0, uint32_t c0, const Box &b1, uint32_t c1)
    {
        float m0[3], e0[3], m1[3], e1[3];
        double w0, k0, w1, k1;
        encode(b0, m0, e0, w0, k0);
        encode(b1, m1, e1, w1, k1);
        float *p = &out->nodes[16 * (size_t)node];
        // pad = A |o - C|^2 + K, or A |m0 + m1 - 2 o|^2 + K (see the header): >= w2 |m - o|^2 + k for both children
        const double u = ldexp(1.0, -24), w2 = std::max(w0, w1), k = std::max(k0, k1);
        const double head = 1.0 + ldexp(1.0, -18);
        float A, K;
        if (!pad_local)
        {
            double g2 = 0;
            for (const float *m : {m0, m1})
            {
                double q = 0;
                for (int a = 0; a < 3; ++a)
                    q += ((double)m[a] - centre[a]) * ((double)m[a] - centre[a]);
                g2 = std::max(g2, q);
            }
            const double c1n = std::fabs(centre[0]) + std::fabs(centre[1]) + std::fabs(centre[2]);
            A = round_up((2.0 * w2 + u) * head), K = round_up((k + 2.0 * w2 * g2 + u * (1.0 + c1n)) * head);
        }
        else
        {
            double g2 = 0, mn2 = 0, M = 0, M0 = 0, M1 = 0;
            for (int a = 0; a < 3; ++a)
            {
                const double h = 0.5 * ((double)m0[a] - (double)m1[a]), mn = 0.5 * ((double)m0[a] + (double)m1[a]);
                g2 += h * h, mn2 += mn * mn;
                M0 = std::max(M0, std::fabs((double)m0[a])), M1 = std::max(M1, std::fabs((double)m1[a]));
            }
            M = M0 + M1;
            // (D + g)^2 <= (1 + eps) D^2 + (1 + 1 / eps) g^2 for every eps > 0: tight at D = g / eps.  eps = g / d_typ makes the
            // pad exact for origins d_typ away (the scene's scale), instead of twice what is needed (eps = 1) everywhere
            const double eps = std::min(1.0, std::max(1.0 / 64.0, std::sqrt(g2) / d_typ));
            const double a_loc = (0.25 * (1.0 + eps) * w2 + u / 8.0) * (1.0 + ldexp(1.0, -10));
            A = round_up(a_loc * head);
            K = round_up((k + (1.0 + 1.0 / eps) * w2 * g2 + u * (0.5 + std::sqrt(mn2)) + a_loc * 3075.0 * 288.0 * u * u * M * M) * head);
        }
        // {m0x m1x m0y m1y} {m0z m1z e0x e1x} {e0y e1y e0z e1z} {A K child0 child1}
        p[0] = m0[0], p[1] = m1[0], p[2] = m0[1], p[3] = m1[1];
        p[4] = m0[2], p[5] = m1[2], p[6] = e0[0], p[7] = e1[0];
        p[8] = e0[1], p[9] = e1[1], p[10] = e0[2], p[11] = e1[2];
        p[12] = A, p[13] = K;
        memcpy(&p[14], &c0, 4);
        memcpy(&p[15], &c1, 4);
    }
};

} // namespace

// Builds the tree over the `na` active spheres (fp32 arrays in active order).  Node 0 is always
// an inner node (the root), even for 0 or 1 spheres.
void r1_build_bvh(uint32_t na, const float *cx, const float *cy, const float *cz, const float *rsq, const double *rbound, int leaf_max,
                  R1Bvh &out)
{
    out.nodes.clear(), out.prims.clear(), out.ids.clear();
    out.max_depth = 0, out.n_leaves = 0;
    if (leaf_max < 1)
        leaf_max = 1;
    if (leaf_max > 14)
        leaf_max = 14; // 7 pairs
    Builder B;
    B.cx = cx, B.cy = cy, B.cz = cz, B.rsq = rsq, B.rbound = rbound;
    B.out = &out;
    B.leaf_max = leaf_max;
    B.sphere.resize(na);
    B.order.resize(na);
    for (uint32_t a = 0; a < na; ++a)
    {
        B.order[a] = a;
        Box &s = B.sphere[a];
        const double c[3] = {cx[a], cy[a], cz[a]};
        // extents cover rbound (>= the radius the exact test reads, r1_bound_radius); the error terms
        // E1 / (2 r) follow the radius the test itself uses, sqrt(radius_sq) — the smaller, safer one
        const double r = rbound[a];
        const double r_test = rsq[a] > 0 ? std::min(r, std::sqrt((double)rsq[a])) : 0.0;
        // degenerate radii: bound sqrt(r^2 + E1) - r through r_floor (AM-GM), see the header
        const double r_floor = 1e-4 * (1.0 + std::fabs(c[0]) + std::fabs(c[1]) + std::fabs(c[2]));
        const double r_eff = std::max(r_test, r_floor);
        for (int k = 0; k < 3; ++k)
            s.lo[k] = c[k] - r, s.hi[k] = c[k] + r, s.clo[k] = s.chi[k] = c[k];
        s.kmax = 1.0 / (2.0 * r_eff);
        s.rmax = r;
        s.floor_pad = r_test < r_floor ? 0.5 * r_floor : 0.0;
    }
    // C: the per-axis median of the centres (the ground sphere's centre, 1000 units below, must not drag it away)
    for (int k = 0; k < 3; ++k)
    {
        const float *src = k == 0 ? cx : (k == 1 ? cy : cz);
        std::vector<float> v(src, src + na);
        double med = 0;
        if (na)
        {
            std::nth_element(v.begin(), v.begin() + na / 2, v.end());
            med = v[na / 2];
        }
        out.centre[k] = (float)med;
        B.centre[k] = (double)out.centre[k]; // exactly the fp32 value the kernel subtracts
    }
    // pad formula of this tree (see the header): per node for scenes of small spheres
    {
        std::vector<double> rr(na), dd(na);
        for (uint32_t a = 0; a < na; ++a)
        {
            rr[a] = B.sphere[a].rmax;
            const double dx = cx[a] - B.centre[0], dy = cy[a] - B.centre[1], dz = cz[a] - B.centre[2];
            dd[a] = dx * dx + dy * dy + dz * dz;
        }
        bool local = false;
        if (na)
        {
            std::nth_element(rr.begin(), rr.begin() + na / 2, rr.end());
            std::nth_element(dd.begin(), dd.begin() + na / 2, dd.end());
            const double r_med = std::max(rr[na / 2], 1e-30), G2 = 4.0 * dd[na / 2];
            const double w2_med = 80.0 * ldexp(1.0, -24) / (2.0 * r_med) + ldexp(1.0, -19);
            local = 4.0 * w2_med * G2 > 0.25 * r_med;
        }
        static const int pad_env = (int)r1_knob("R1_BVH_PAD_LOCAL", -1); // tuning experiments
        if (pad_env >= 0)
            local = pad_env != 0;
        B.pad_local = local;
        B.d_typ = na ? std::max(std::sqrt(4.0 * dd[na / 2]), 1e-30) : 1.0;
        out.pad_local = local ? 1 : 0;
    }
    // the root is node 0
    out.nodes.resize(16);
    Box empty;
    empty.clear();
    auto fill_root_with = [&](const Box &b0, uint32_t c0, const Box *b1, uint32_t c1) {
        if (b1)
        {
            B.fill(0, b0, c0, *b1, c1);
            return;
        }
        // one child only: the other never passes (half extent -inf) and is an empty leaf anyway
        B.fill(0, b0, c0, b0, c1);
        float *p = &out.nodes[0];
        p[7] = p[9] = p[11] = -INFINITY;
    };
    const uint32_t EMPTY_LEAF = Builder::LEAF; // count 0
    if (na == 0)
    {
        Box z;
        z.clear();
        for (int k = 0; k < 3; ++k)
            z.lo[k] = z.hi[k] = z.clo[k] = z.chi[k] = 0;
        fill_root_with(z, EMPTY_LEAF, nullptr, EMPTY_LEAF);
        float *p = &out.nodes[0];
        p[6] = p[8] = p[10] = -INFINITY;
        out.max_depth = 1;
    }
    else if (na <= (uint32_t)leaf_max)
    {
        const Box all = B.bound(0, na);
        const uint32_t leaf = B.make_leaf(0, na);
        fill_root_with(all, leaf, nullptr, EMPTY_LEAF);
        out.max_depth = 1;
    }
    else
    {
        // build() allocates its node first: make the top call land on node 0
        out.nodes.clear();
        const Box all = B.bound(0, na);
        const uint32_t root = B.build(0, na, 0, all);
        (void)root; // == 0
    }
    // Node order: the top of the tree breadth first (the first R1_BVH_TOP_NODES nodes = the levels every walk starts
    // with, which the big-scene kernels keep in LDS), the rest depth first below them (a parent next to its first child,
    // as build() allocates them).  Child references are rewritten; node 0 stays the root.
    {
        const uint32_t nn = (uint32_t)(out.nodes.size() / 16);
        static const int top_env = (int)r1_knob("R1_BVH_TOP", R1_BVH_TOP_NODES); // tuning experiments
        const uint32_t top = (uint32_t)std::max(0, top_env);
        if (nn > 1 && top > 1)
        {
            auto child = [&](uint32_t n, int c) {
                uint32_t r;
                memcpy(&r, &out.nodes[16 * (size_t)n + 14 + c], 4);
                return r;
            };
            std::vector<uint32_t> order_new; // old index of the node at each new position
            order_new.reserve(nn);
            std::vector<uint32_t> frontier(1, 0u);
            size_t head = 0;
            while (head < frontier.size() && frontier.size() < top) // breadth first until `top` nodes are known
            {
                const uint32_t n = frontier[head++];
                for (int c = 0; c < 2; ++c)
                    if (!(child(n, c) & Builder::LEAF))
                        frontier.push_back(child(n, c));
            }
            // frontier[0 .. head) are expanded, frontier[head ..) are the roots of the remaining subtrees
            std::vector<char> placed(nn, 0);
            for (uint32_t n : frontier)
                order_new.push_back(n), placed[n] = 1;
            for (size_t f = head; f < frontier.size(); ++f) // depth first below every unexpanded top node
            {
                std::vector<uint32_t> stack;
                for (int c = 1; c >= 0; --c)
                    if (!(child(frontier[f], c) & Builder::LEAF))
                        stack.push_back(child(frontier[f], c));
                while (!stack.empty())
                {
                    const uint32_t n = stack.back();
                    stack.pop_back();
                    if (placed[n])
                        continue;
                    order_new.push_back(n), placed[n] = 1;
                    for (int c = 1; c >= 0; --c)
                        if (!(child(n, c) & Builder::LEAF))
                            stack.push_back(child(n, c));
                }
            }
            if (order_new.size() == nn)
            {
                std::vector<uint32_t> new_of(nn);
                for (uint32_t i = 0; i < nn; ++i)
                    new_of[order_new[i]] = i;
                std::vector<float> moved(out.nodes.size());
                for (uint32_t i = 0; i < nn; ++i)
                {
                    memcpy(&moved[16 * (size_t)i], &out.nodes[16 * (size_t)order_new[i]], 64);
                    for (int c = 0; c < 2; ++c)
                    {
                        uint32_t r;
                        memcpy(&r, &moved[16 * (size_t)i + 14 + c], 4);
                        if (!(r & Builder::LEAF))
                        {
                            r = new_of[r];
                            memcpy(&moved[16 * (size_t)i + 14 + c], &r, 4);
                        }
                    }
                }
                out.nodes.swap(moved);
            }
        }
    }
    // Trees whose pad is measured from the scene's centre: ONE A for the whole tree (the largest of the nodes': pad only grows) and
    // every node's K folded into its half extents, e' = e + K rounded up — b = e |1/d| + (A R2 + K) |1/d| = e' |1/d| + (A R2) |1/d| —
    // so that the kernels which keep the table in LDS evaluate A R2 |1/d| once per RAY instead of a fused multiply-add and three
    // products per node visit (bvh_advance; 51 -> 47 VALU instructions per visit).  The same number of fp32 roundings as before on the
    // way to b (product, product, fused multiply-add), covered by the same 2^-18 headroom on A and K.  Nodes keep {A, 0} in the pad
    // slots: the kernels that read the pad per node (table in global memory) and the CPU restatement of the visit rule see the same rule.
    if (!B.pad_local)
    {
        const size_t nn = out.nodes.size() / 16;
        float a_max = 0.0f;
        for (size_t n = 0; n < nn; ++n)
            a_max = std::max(a_max, out.nodes[16 * n + 12]);
        for (size_t n = 0; n < nn; ++n)
        {
            float *q = &out.nodes[16 * n];
            const double K = q[13];
            for (int j = 6; j < 12; ++j)
                if (std::isfinite(q[j])) // (-inf: a child that never passes)
                    q[j] = round_up((double)q[j] + K);
            q[12] = a_max, q[13] = 0.0f;
        }
    }
    // The root of the reference's scenes is [one leaf of outliers: the ground + the three big balls | the lattice] (the peeling above),
    // and EVERY ray tests that leaf.  The tree kernels take this step out of the walk's divergent loops: a ray that starts its walk tests
    // the root's leaf child (<= 2 pairs) and then the box of the other child in straight-line code, together with the other rays of the
    // wave that start in the same iteration, and the walk begins at the other child (bvh_advance).  root_leaf says whether the root has
    // that shape and which child is the leaf.
    out.root_leaf = 0;
    if (out.nodes.size() >= 16)
    {
        uint32_t c[2];
        memcpy(c, &out.nodes[14], 8);
        for (int k = 0; k < 2; ++k)
        {
            const uint32_t pairs = (c[k] >> 28) & 7u;
            if ((c[k] & Builder::LEAF) && !(c[1 - k] & Builder::LEAF) && pairs >= 1 && pairs <= 2)
                out.root_leaf = k + 1;
        }
    }
    // keep the tables non-empty for the uploader
    if (out.prims.empty())
        out.prims.assign(8, 0.0f), out.ids.assign(2, 0xFFFFFFFFu);
}

// The hittable ("active") spheres of a caller's scene, in scene order: inv_radius != 0
// (rayweek1.cpp:291).  A sphere with a non-finite centre or radius_sq can never be hit by the
// reference's arithmetic (NaN/inf discriminant or roots fail every compare, rayweek1.cpp:204,
// :297-309) and is dropped like a placeholder, which also keeps such values out of the builders.
// One helper for r1_set_scene and r1_bvh_describe, so both see the same spheres.
extern "C" void r1_set_error(const char *fmt, ...);
int r1_active_spheres(const r1_scene *s, std::vector<uint32_t> &active_to_scene)
{
    active_to_scene.clear();
    for (uint32_t i = 0; i < s->count; ++i)
    {
        if (std::isnan(s->inv_radius[i]))
        {
            r1_set_error("scene: sphere %u has inv_radius NaN", i);
            return R1_EINVAL;
        }
        if (s->inv_radius[i] == 0)
            continue;
        if (!std::isfinite(s->center_x[i]) || !std::isfinite(s->center_y[i]) || !std::isfinite(s->center_z[i]) ||
            !std::isfinite(s->radius_sq[i]))
            continue;
        active_to_scene.push_back(i);
    }
    return R1_OK;
}

// Radius every conservative bound (group bounding spheres, tree boxes) must cover.  The hit test
// reads radius_sq only (rayweek1.cpp:198); SphereSOA::add stores radius_sq = r*r and inv_radius =
// 1/r of the same r (soa_sphere.cpp:70-85), but a C-ABI caller may hand over arrays that disagree
// (or a negative inv_radius): taking the larger of the two keeps every bound conservative for
// whatever the exact test can accept instead of silently dropping hits.
double r1_bound_radius(float radius_sq, float inv_radius)
{
    const double from_sq = radius_sq > 0 ? std::sqrt((double)radius_sq) : 0.0;
    const double from_inv = std::isfinite(inv_radius) && inv_radius != 0 ? 1.0 / std::fabs((double)inv_radius) : 0.0;
    return std::max(from_sq, from_inv);
}

// Host-only view of the index (include/rays1.h): what r1_set_scene would build for this scene.
extern "C" int r1_bvh_describe(const r1_scene *s, int32_t leaf_max, r1_bvh_info *info, float *nodes_out, size_t nodes_cap, uint32_t *ids_out,
                               size_t ids_cap)
{
    if (!s || !info || !s->center_x || !s->center_y || !s->center_z || !s->radius_sq || !s->inv_radius)
        return R1_EINVAL;
    std::vector<float> x, y, z, r;
    std::vector<double> rb;
    std::vector<uint32_t> scene_index;
    if (r1_active_spheres(s, scene_index) != R1_OK)
        return R1_EINVAL;
    for (uint32_t i : scene_index)
    {
        x.push_back(s->center_x[i]), y.push_back(s->center_y[i]), z.push_back(s->center_z[i]), r.push_back(s->radius_sq[i]);
        rb.push_back(r1_bound_radius(s->radius_sq[i], s->inv_radius[i]));
    }
    const uint32_t na = (uint32_t)x.size();
    if (na == 0)
        x.push_back(0), y.push_back(0), z.push_back(0), r.push_back(0), rb.push_back(0);
    R1Bvh b;
    r1_build_bvh(na, x.data(), y.data(), z.data(), r.data(), rb.data(),
                 leaf_max > 0 ? leaf_max : (na > R1_MAX_ACTIVE_10BIT ? 2 * R1_BVH_LEAF : R1_BVH_LEAF), b);
    info->nodes = (int32_t)(b.nodes.size() / 16);
    info->leaves = (int32_t)b.n_leaves;
    info->depth = b.max_depth;
    info->stack_entries = R1_BVH_STACK;
    info->spheres = (int32_t)na;
    for (int k = 0; k < 3; ++k)
        info->centre[k] = b.centre[k];
    info->pad_local = b.pad_local;
    info->root_leaf = b.root_leaf;
    if (nodes_out)
    {
        if (nodes_cap < b.nodes.size())
            return R1_ELIMIT;
        memcpy(nodes_out, b.nodes.data(), b.nodes.size() * 4);
    }
    info->pairs = (int32_t)(b.ids.size() / 2);
    if (ids_out)
    {
        if (ids_cap < b.ids.size())
            return R1_ELIMIT;
        for (size_t i = 0; i < b.ids.size(); ++i) // leaf slot (2 per pair) -> index into the caller's scene arrays
            ids_out[i] = b.ids[i] == 0xFFFFFFFFu || na == 0 ? 0xFFFFFFFFu : scene_index[b.ids[i]];
    }
    return R1_OK;
}
