// trace_sim.h -- TEST INFRASTRUCTURE (debug harness only): models how the lanes of a 64-wide wave would spend
// their vector-ALU cycles in the traversal kernel under different loop shapes / ray orders, from the real ray
// queues of a render.  Enabled with RDR_TRACE_SIM=1; prints a summary at exit.  Costs are instruction counts read
// off the gfx950 ISA of trace.hip: inner step ~110, triangle test ~190, ray set-up ~150.
#pragma once
#include "bvh.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tracesim {

struct Seg { int inner; int tris; };            // inner steps, then one leaf with `tris` triangle tests (0 = none: end)
using Path = std::vector<Seg>;

template <bool ANY>
inline Path record(const rt::BvhD &bvh, const float o[3], const float d[3], float tnear, float tfar) {
    using namespace rt;
    Path path;
    Hit best{tfar, -1, -1};
    const float inv[3] = {1.f / d[0], 1.f / d[1], 1.f / d[2]};
    int stack[kTraverseStack]; int sp = 0; float tn;
    const Node &root = bvh.nodes[0];
    if (!ray_box(o, inv, tnear, tfar, root.lo, root.hi, &tn)) return path;
    int ca = root.a, cb = root.b;
    for (;;) {
        int inner = 0;
        while (cb == 0) {
            ++inner;
            const Node l = bvh.nodes[ca], r = bvh.nodes[ca + 1];
            const float lim = best.shape < 0 ? tfar : best.t * 1.0000004f + 1e-30f;
            float tl, tr;
            const bool hl = ray_box(o, inv, tnear, lim, l.lo, l.hi, &tl);
            const bool hr = ray_box(o, inv, tnear, lim, r.lo, r.hi, &tr);
            if (hl || hr) {
                const bool lf = !hr || (hl && !(tr < tl));
                if (hl && hr) stack[sp++] = lf ? ca + 1 : ca;
                ca = lf ? l.a : r.a; cb = lf ? l.b : r.b;
            } else if (sp == 0) cb = -1;
            else { const Node &nx = bvh.nodes[stack[--sp]]; ca = nx.a; cb = nx.b; }
        }
        if (cb < 0) { if (inner) path.push_back(Seg{inner, 0}); break; }
        path.push_back(Seg{inner, cb});
        bool stop = false;
        for (int k = 0; k < cb; ++k) {
            const int slot = ca + k;
            const float *t = bvh.tris + 9 * slot;
            float th;
            if (ray_triangle(o, d, tnear, tfar, t, t + 3, t + 6, &th)) {
                const int s = bvh.ids[2 * slot], p = bvh.ids[2 * slot + 1];
                if (ANY) { stop = true; break; }
                if (closer(th, s, p, best)) best = Hit{th, s, p};
            }
        }
        if (stop || sp == 0) break;
        const Node &nx = bvh.nodes[stack[--sp]]; ca = nx.a; cb = nx.b;
    }
    return path;
}

constexpr double Ci = 110, Ct = 190, Cr = 150;

struct Tot { double rep[8] = {0}; double bud[12] = {0}; double vw[6] = {0, 0, 0, 0, 0, 0}; double multi_s[3] = {0, 0, 0}, multi_d[3] = {0, 0, 0}; double vote[4] = {0, 0, 0, 0}, vote_sorted[4] = {0, 0, 0, 0}; double spec[12] = {0}; double useful = 0, ifif = 0, whilewhile = 0, ww_sorted = 0, refill8 = 0, refill24 = 0, refill_sorted = 0; long rays = 0, live = 0; };
inline Tot &tot(bool any) { static Tot t[2]; return t[any ? 1 : 0]; }

inline double useful_of(const Path &p) { double c = 0; for (const Seg &s : p) c += Ci * s.inner + Ct * s.tris; return c; }

// one wave, one ray per lane, "if-if": every iteration each lane does its next op (inner step or whole leaf)
inline double sim_ifif(const Path *const *lanes, int n) {
    std::vector<size_t> seg(n, 0); std::vector<int> done_inner(n, 0);
    double cost = 0;
    for (;;) {
        bool any_i = false; int max_t = 0; bool alive = false;
        for (int l = 0; l < n; ++l) {
            const Path &p = *lanes[l];
            if (seg[l] >= p.size()) continue;
            alive = true;
            const Seg &s = p[seg[l]];
            if (done_inner[l] < s.inner) { any_i = true; ++done_inner[l]; if (done_inner[l] == s.inner && s.tris == 0) { ++seg[l]; done_inner[l] = 0; } }
            else { max_t = std::max(max_t, s.tris); ++seg[l]; done_inner[l] = 0; }
        }
        if (!alive) break;
        cost += (any_i ? Ci : 0) + Ct * max_t;
    }
    return cost;
}
inline double sim_ww(const Path *const *lanes, int n) {
    std::vector<size_t> seg(n, 0);
    double cost = 0;
    for (;;) {
        int max_i = 0, max_t = 0; bool alive = false;
        for (int l = 0; l < n; ++l) {
            const Path &p = *lanes[l];
            if (seg[l] >= p.size()) continue;
            alive = true;
            max_i = std::max(max_i, p[seg[l]].inner); max_t = std::max(max_t, p[seg[l]].tris);
            ++seg[l];
        }
        if (!alive) break;
        cost += Ci * max_i + Ct * max_t;
    }
    return cost;
}
// "if-if" flat loop with several rays per lane: a wave owns 64 * k consecutive rays; static: lane l takes rays l, l + 64, ...
// of them in turn; dynamic: a lane that finishes takes the next unclaimed ray of the wave's block.  A lane that starts a ray
// pays the set-up as one more kind of step of the iteration.
inline double sim_multi(const Path *const *rays, int n, int k, bool dynamic) {
    const int lanes = 64;
    std::vector<int> cur(lanes, -1), nextj(lanes, 0);
    std::vector<size_t> seg(lanes, 0); std::vector<int> done_inner(lanes, 0);
    int pool = 0;                                   // dynamic: next unclaimed ray
    double cost = 0;
    auto take = [&](int l) -> bool {
        for (;;) {
            int r;
            if (dynamic) { if (pool >= n) return false; r = pool++; }
            else { r = l + lanes * nextj[l]; if (nextj[l] >= k || r >= n) return false; ++nextj[l]; }
            cur[l] = r; seg[l] = 0; done_inner[l] = 0;
            return true;
        }
    };
    std::vector<char> finished(lanes, 0);
    for (;;) {
        bool any_setup = false, any_i = false, alive = false; int max_t = 0;
        for (int l = 0; l < lanes; ++l) {
            if (finished[l]) continue;
            if (cur[l] < 0 || seg[l] >= rays[cur[l]]->size()) {
                if (!take(l)) { finished[l] = 1; continue; }
                any_setup = true; alive = true;
                continue;                           // the set-up is this lane's step of the iteration
            }
            alive = true;
            const Path &p = *rays[cur[l]];
            const Seg &sg = p[seg[l]];
            if (done_inner[l] < sg.inner) { any_i = true; ++done_inner[l]; if (done_inner[l] == sg.inner && sg.tris == 0) { ++seg[l]; done_inner[l] = 0; } }
            else { max_t = std::max(max_t, sg.tris); ++seg[l]; done_inner[l] = 0; }
        }
        if (!alive) break;
        cost += (any_setup ? Cr : 0) + (any_i ? Ci : 0) + Ct * max_t;
    }
    return cost;
}

// persistent wave over a whole queue, while-while, lanes refilled when >= `idle_min` of them are idle
inline double sim_refill(const std::vector<const Path *> &q, int idle_min) {
    size_t next = 0;
    const Path *cur[64]; size_t seg[64];
    for (int l = 0; l < 64; ++l) cur[l] = nullptr;
    double cost = 0;
    for (;;) {
        int idle = 0;
        for (int l = 0; l < 64; ++l) if (!cur[l]) ++idle;
        if (idle == 64 && next >= q.size()) break;
        if ((idle >= idle_min || idle == 64) && next < q.size()) {
            for (int l = 0; l < 64 && next < q.size(); ++l) if (!cur[l]) { cur[l] = q[next++]; seg[l] = 0; if (cur[l]->empty()) cur[l] = nullptr; }
            cost += Cr;
        }
        int max_i = 0, max_t = 0; bool alive = false;
        for (int l = 0; l < 64; ++l) {
            if (!cur[l]) continue;
            alive = true;
            const Seg &s = (*cur[l])[seg[l]];
            max_i = std::max(max_i, s.inner); max_t = std::max(max_t, s.tris);
            if (++seg[l] >= cur[l]->size()) cur[l] = nullptr;
        }
        if (alive) cost += Ci * max_i + Ct * max_t;
    }
    return cost;
}

// persistent wave with refill (>= idle_min idle lanes), SPECULATIVE leaves: a lane that reaches a leaf puts it in a pocket
// (`pockets` of them) and goes on with its walk; it waits only when it reaches a leaf with every pocket full or has nothing left
// to walk.  The pockets of the whole wave are tested together (a loop over the largest pocket content) when >= pend_min lanes
// hold one, when >= wait_min lanes wait, or when nobody can walk.  `vote`: instructions per iteration for the ballots.
inline double sim_spec(const std::vector<const Path *> &q, int idle_min, int pockets, int pend_min, int wait_min, double vote) {
    struct Ev { bool leaf; int tris; };
    size_t next = 0;
    std::vector<Ev> ev[64]; size_t at[64]; int pocket[64]; int pocket_n[64]; bool has[64];
    for (int l = 0; l < 64; ++l) { has[l] = false; pocket[l] = 0; pocket_n[l] = 0; at[l] = 0; }
    double cost = 0;
    auto load = [&](int l, const Path *p) {
        ev[l].clear(); at[l] = 0; pocket[l] = 0; pocket_n[l] = 0;
        for (const Seg &s : *p) { for (int i = 0; i < s.inner; ++i) ev[l].push_back(Ev{false, 0}); if (s.tris > 0) ev[l].push_back(Ev{true, s.tris}); }
        has[l] = !ev[l].empty();
    };
    for (;;) {
        int idle = 0;
        for (int l = 0; l < 64; ++l) if (!has[l]) ++idle;
        if (idle == 64 && next >= q.size()) break;
        if ((idle >= idle_min || idle == 64) && next < q.size()) {
            for (int l = 0; l < 64 && next < q.size(); ++l) if (!has[l]) load(l, q[next++]);
            cost += Cr;
        }
        // who can walk: has events left and (next event is not a leaf or a pocket is free)
        int walkers = 0, pend = 0, waiting = 0;
        for (int l = 0; l < 64; ++l) {
            if (!has[l]) continue;
            const bool more = at[l] < ev[l].size();
            const bool can = more && (!ev[l][at[l]].leaf || pocket_n[l] < pockets);
            if (can) ++walkers; else ++waiting;
            if (pocket_n[l] > 0) ++pend;
        }
        cost += vote;
        const bool flush = pend > 0 && (pend >= pend_min || waiting >= wait_min || walkers == 0);
        if (flush) {
            int max_t = 0;
            for (int l = 0; l < 64; ++l) if (has[l]) { max_t = std::max(max_t, pocket[l]); pocket[l] = 0; pocket_n[l] = 0; }
            cost += Ct * max_t;
            for (int l = 0; l < 64; ++l) if (has[l] && at[l] >= ev[l].size()) has[l] = false;
            continue;
        }
        if (walkers > 0) {
            cost += Ci;
            for (int l = 0; l < 64; ++l) {
                if (!has[l] || at[l] >= ev[l].size()) continue;
                const Ev &e = ev[l][at[l]];
                if (e.leaf) { if (pocket_n[l] < pockets) { pocket[l] += e.tris; ++pocket_n[l]; ++at[l]; } }
                else ++at[l];
                if (at[l] >= ev[l].size() && pocket_n[l] == 0) has[l] = false;
            }
        }
    }
    return cost;
}

// persistent wave, lanes refilled when >= idle_min are idle; inner steps run one at a time for the lanes that are
// not parked on a leaf, and the parked leaves are tested (one triangle per lane per step) as soon as at least
// `park_min` lanes are parked or nobody can do an inner step
inline double sim_vote(const std::vector<const Path *> &q, int idle_min, int park_min) {
    size_t next = 0;
    const Path *cur[64]; size_t seg[64]; int inner_left[64], tris_left[64];
    for (int l = 0; l < 64; ++l) cur[l] = nullptr;
    double cost = 0;
    auto load_seg = [&](int l) {
        while (cur[l]) {
            if (seg[l] >= cur[l]->size()) { cur[l] = nullptr; break; }
            inner_left[l] = (*cur[l])[seg[l]].inner; tris_left[l] = (*cur[l])[seg[l]].tris;
            if (inner_left[l] == 0 && tris_left[l] == 0) { ++seg[l]; continue; }
            break;
        }
    };
    for (;;) {
        int idle = 0;
        for (int l = 0; l < 64; ++l) if (!cur[l]) ++idle;
        if (idle == 64 && next >= q.size()) break;
        if ((idle >= idle_min || idle == 64) && next < q.size()) {
            for (int l = 0; l < 64 && next < q.size(); ++l) if (!cur[l]) { cur[l] = q[next++]; seg[l] = 0; load_seg(l); }
            cost += Cr;
        }
        int want_inner = 0, parked = 0;
        for (int l = 0; l < 64; ++l) if (cur[l]) { if (inner_left[l] > 0) ++want_inner; else ++parked; }
        if (want_inner == 0 && parked == 0) continue;
        if (parked >= park_min || want_inner == 0) {
            cost += Ct;
            for (int l = 0; l < 64; ++l) if (cur[l] && inner_left[l] == 0) { if (--tris_left[l] == 0) { ++seg[l]; load_seg(l); } }
        } else {
            cost += Ci;
            for (int l = 0; l < 64; ++l) if (cur[l] && inner_left[l] > 0) { --inner_left[l]; if (inner_left[l] == 0 && tris_left[l] == 0) { ++seg[l]; load_seg(l); } }
        }
    }
    return cost;
}

// one wave, one ray per lane, NO refill: lanes park on leaves; an iteration is either one inner step (for the lanes that are
// not parked) or one triangle test per parked lane, by vote: triangles when >= park_min lanes are parked or nobody can do an
// inner step.  `overhead`: extra instructions per iteration for the vote.
inline double sim_vote_wave(const Path *const *lanes, int n, int park_min, double overhead) {
    std::vector<size_t> seg(n, 0); std::vector<int> inner_left(n, 0), tris_left(n, 0); std::vector<char> live(n, 0);
    auto load_seg = [&](int l) {
        for (;;) {
            if (seg[l] >= lanes[l]->size()) { live[l] = 0; return; }
            inner_left[l] = (*lanes[l])[seg[l]].inner; tris_left[l] = (*lanes[l])[seg[l]].tris;
            if (inner_left[l] == 0 && tris_left[l] == 0) { ++seg[l]; continue; }
            return;
        }
    };
    for (int l = 0; l < n; ++l) { live[l] = 1; load_seg(l); }
    double cost = 0;
    for (;;) {
        int want_inner = 0, parked = 0;
        for (int l = 0; l < n; ++l) if (live[l]) { if (inner_left[l] > 0) ++want_inner; else ++parked; }
        if (want_inner == 0 && parked == 0) break;
        cost += overhead;
        if (parked >= park_min || want_inner == 0) {
            cost += Ct;
            for (int l = 0; l < n; ++l) if (live[l] && inner_left[l] == 0) { if (--tris_left[l] == 0) { ++seg[l]; load_seg(l); } }
        } else {
            cost += Ci;
            for (int l = 0; l < n; ++l) if (live[l] && inner_left[l] > 0) { --inner_left[l]; if (inner_left[l] == 0 && tris_left[l] == 0) { ++seg[l]; load_seg(l); } }
        }
    }
    return cost;
}

// Work budget + continuation queue: pass k runs every wave (64 consecutive queue entries, if-if body; vote = true: parked
// lanes + vote instead) for at most budget[k] iterations; the lanes that are not finished by then are appended, in order, to
// the queue of the next pass (compacted: long rays end up together, in full waves), where they resume.  `hand_off`: instructions
// per wave and pass for saving / restoring the state of the continuing lanes.
struct LaneState { const Path *p; size_t seg; int done_inner; int tris_done; };
inline double sim_budget(const std::vector<const Path *> &q0, const int *budget, int passes, bool vote, int park_min, double hand_off) {
    std::vector<LaneState> q;
    for (const Path *p : q0) q.push_back(LaneState{p, 0, 0, 0});
    double cost = 0;
    for (int pass = 0; pass < passes && !q.empty(); ++pass) {
        std::vector<LaneState> next;
        const int B = pass == passes - 1 ? 1 << 30 : budget[pass];
        for (size_t w = 0; w < q.size(); w += 64) {
            const int n = (int)std::min<size_t>(64, q.size() - w);
            LaneState *L = q.data() + w;
            bool handed = false;
            for (int it = 0; it < B; ++it) {
                bool any_i = false, alive = false; int max_t = 0, want_inner = 0, parked = 0;
                for (int l = 0; l < n; ++l) {
                    if (L[l].seg >= L[l].p->size()) continue;
                    alive = true;
                    const Seg &sg = (*L[l].p)[L[l].seg];
                    if (L[l].done_inner < sg.inner) ++want_inner; else ++parked;
                }
                if (!alive) break;
                if (!vote) {
                    for (int l = 0; l < n; ++l) {
                        if (L[l].seg >= L[l].p->size()) continue;
                        const Seg &sg = (*L[l].p)[L[l].seg];
                        if (L[l].done_inner < sg.inner) { any_i = true; ++L[l].done_inner; if (L[l].done_inner == sg.inner && sg.tris == 0) { ++L[l].seg; L[l].done_inner = 0; } }
                        else { max_t = std::max(max_t, sg.tris); ++L[l].seg; L[l].done_inner = 0; }
                    }
                    cost += (any_i ? Ci : 0) + Ct * max_t;
                } else {
                    cost += 8;
                    const bool do_tris = parked >= park_min || want_inner == 0;
                    cost += do_tris ? Ct : Ci;
                    for (int l = 0; l < n; ++l) {
                        if (L[l].seg >= L[l].p->size()) continue;
                        const Seg &sg = (*L[l].p)[L[l].seg];
                        const bool inner = L[l].done_inner < sg.inner;
                        if (do_tris && !inner) { if (++L[l].tris_done >= sg.tris) { ++L[l].seg; L[l].done_inner = 0; L[l].tris_done = 0; } }
                        else if (!do_tris && inner) { ++L[l].done_inner; if (L[l].done_inner == sg.inner && sg.tris == 0) { ++L[l].seg; L[l].done_inner = 0; } }
                    }
                }
            }
            for (int l = 0; l < n; ++l) if (L[l].seg < L[l].p->size()) { next.push_back(L[l]); handed = true; }
            if (handed) cost += hand_off;
            if (pass > 0) cost += hand_off;
        }
        q.swap(next);
    }
    return cost;
}

// Workgroup-local re-packing: a workgroup of `wg` consecutive rays (wg / 64 waves, if-if body); every `every` iterations the
// rays that are still alive are packed densely into the first waves of the workgroup (waves without rays issue nothing).
// `repack_cost`: instructions per live wave and re-pack.
inline double sim_repack(const std::vector<const Path *> &q0, int wg, int every, double repack_cost) {
    double cost = 0;
    for (size_t base = 0; base < q0.size(); base += wg) {
        std::vector<LaneState> L;
        for (size_t i = base; i < std::min(q0.size(), base + (size_t)wg); ++i) L.push_back(LaneState{q0[i], 0, 0, 0});
        for (;;) {
            // drop finished rays (re-pack), keep order
            std::vector<LaneState> live;
            for (const LaneState &l : L) if (l.seg < l.p->size()) live.push_back(l);
            L.swap(live);
            if (L.empty()) break;
            const int waves = ((int)L.size() + 63) / 64;
            cost += repack_cost * waves;
            for (int w = 0; w < waves; ++w) {
                LaneState *W = L.data() + (size_t)w * 64;
                const int n = std::min(64, (int)L.size() - w * 64);
                for (int it = 0; it < every; ++it) {
                    bool any_i = false, alive = false; int max_t = 0;
                    for (int l = 0; l < n; ++l) {
                        if (W[l].seg >= W[l].p->size()) continue;
                        alive = true;
                        const Seg &sg = (*W[l].p)[W[l].seg];
                        if (W[l].done_inner < sg.inner) { any_i = true; ++W[l].done_inner; if (W[l].done_inner == sg.inner && sg.tris == 0) { ++W[l].seg; W[l].done_inner = 0; } }
                        else { max_t = std::max(max_t, sg.tris); ++W[l].seg; W[l].done_inner = 0; }
                    }
                    if (!alive) break;
                    cost += (any_i ? Ci : 0) + Ct * max_t;
                }
            }
        }
    }
    return cost;
}

// RDR_TRACE_SIM_KEY: "<bits per axis of the origin grid><o|n: with / without the direction octant><f|l: octant first / last>"
inline unsigned sort_key(const rt::BvhD &bvh, const rt::RayRec &r) {
    static const char *mode = std::getenv("RDR_TRACE_SIM_KEY") ? std::getenv("RDR_TRACE_SIM_KEY") : "4ol";
    const int bits = mode[0] - '0';
    const bool with_oct = mode[1] == 'o', oct_first = mode[2] == 'f';
    const rt::Node &root = bvh.nodes[0];
    unsigned key = 0;
    const float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
    unsigned c[3];
    for (int k = 0; k < 3; ++k) {
        float u = (o[k] - root.lo[k]) / (root.hi[k] - root.lo[k]);
        c[k] = (unsigned)std::min((float)((1 << bits) - 1), std::max(0.f, u * (float)(1 << bits)));
    }
    unsigned oct = 0;
    for (int k = 0; k < 3; ++k) oct = (oct << 1) | (d[k] < 0 ? 1u : 0u);
    if (with_oct && oct_first) key = oct;
    for (int b = bits - 1; b >= 0; --b) for (int k = 0; k < 3; ++k) key = (key << 1) | ((c[k] >> b) & 1u);
    if (with_oct && !oct_first) key = (key << 3) | oct;
    return key;
}

inline void report() {
    for (int a = 0; a < 2; ++a) {
        const Tot &t = tot(a != 0);
        if (!t.rays) continue;
        std::fprintf(stderr, "[trace_sim] %s: %ld rays (%ld live), useful %.0f/ray; wave cost per ray and lane efficiency:\n", a ? "any-hit" : "closest", t.rays, t.live, t.useful / t.rays);
        auto line = [&](const char *name, double c) { std::fprintf(stderr, "[trace_sim]   %-28s %8.0f  eff %.2f\n", name, c * 64 / t.rays, t.useful / (c * 64)); };
        line("if-if", t.ifif);
        { const int ks[3] = {2, 4, 8}; for (int i = 0; i < 3; ++i) { char nm[64]; std::snprintf(nm, sizeof nm, "if-if, %d rays/lane static", ks[i]); line(nm, t.multi_s[i]); std::snprintf(nm, sizeof nm, "if-if, %d rays/lane dynamic", ks[i]); line(nm, t.multi_d[i]); } }
        line("while-while", t.whilewhile); line("while-while, sorted rays", t.ww_sorted);
        const int parks[4] = {8, 16, 32, 48};
        for (int i = 0; i < 4; ++i) { char nm[64]; std::snprintf(nm, sizeof nm, "vote park>=%d", parks[i]); line(nm, t.vote[i]); }
        for (int i = 0; i < 4; ++i) { char nm[64]; std::snprintf(nm, sizeof nm, "vote park>=%d, sorted", parks[i]); line(nm, t.vote_sorted[i]); }
        { const int pk[6] = {1, 4, 8, 16, 24, 32}; for (int i = 0; i < 6; ++i) { char nm[64]; std::snprintf(nm, sizeof nm, "vote, no refill, park>=%d", pk[i]); line(nm, t.vw[i]); } }
        { const char *nm[12] = {"budget 16,inf", "budget 24,inf", "budget 32,inf", "budget 16,32,inf", "budget 24,48,inf", "budget 12,24,48,inf",
                                "budget 24,inf vote8", "budget 32,inf vote8", "budget 24,48,inf vote8", "budget 32,64,inf vote8", "budget 32,64,inf vote16", "budget 48,96,inf vote8"};
          for (int i = 0; i < 12; ++i) line(nm[i], t.bud[i]); }
        { const char *nm[8] = {"repack wg 256 every 8", "repack wg 512 every 8", "repack wg 1024 every 8", "repack wg 1024 every 4", "repack wg 1024 every 16", "repack wg 1024 every 12", "repack wg 2048 every 8", "repack wg 1024 every 8 (free)"};
          for (int i = 0; i < 8; ++i) line(nm[i], t.rep[i]); }
        { const char *nm[12] = {"spec 1 pocket pend16 wait8", "spec 1 pocket pend24 wait12", "spec 1 pocket pend32 wait16", "spec 2 pockets pend16 wait8", "spec 2 pockets pend24 wait12", "spec 2 pockets pend32 wait16",
                                "spec 2 pockets pend32 wait24", "spec 2 pockets pend48 wait24", "spec 1 pocket pend8 wait4", "spec 2 pockets pend24 wait12 idle8", "spec 1 pocket pend1 wait1 (=if-if refill)", "spec 3 pockets pend32 wait16"};
          for (int i = 0; i < 12; ++i) line(nm[i], t.spec[i]); }
        line("refill (>= 8 idle)", t.refill8); line("refill (>= 24 idle)", t.refill24); line("refill (>= 8), sorted", t.refill_sorted);
    }
}

inline void launch(const rt::BvhD &bvh, const rt::RayRec *rays, int n, bool any) {
    static bool registered = false;
    if (!registered) { registered = true; std::atexit(report); }
    std::vector<Path> paths(n);
    Tot &t = tot(any);
    for (int i = 0; i < n; ++i) {
        const rt::RayRec &r = rays[i];
        if (r.tmax < 0.f) continue;
        float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
        paths[i] = any ? record<true>(bvh, o, d, r.tmin, r.tmax) : record<false>(bvh, o, d, r.tmin, r.tmax);
        t.useful += useful_of(paths[i]);
        ++t.live;
    }
    t.rays += n;
    std::vector<const Path *> q(n);
    for (int i = 0; i < n; ++i) q[i] = &paths[i];
    for (int w = 0; w < n; w += 64) { int m = std::min(64, n - w); t.ifif += sim_ifif(q.data() + w, m); t.whilewhile += sim_ww(q.data() + w, m); }
    { const int pk[6] = {1, 4, 8, 16, 24, 32}; for (int i = 0; i < 6; ++i) for (int w = 0; w < n; w += 64) { int m = std::min(64, n - w); t.vw[i] += sim_vote_wave(q.data() + w, m, pk[i], 8); } }
    { const int b[12][3] = {{16, 0, 0}, {24, 0, 0}, {32, 0, 0}, {16, 32, 0}, {24, 48, 0}, {12, 24, 48}, {24, 0, 0}, {32, 0, 0}, {24, 48, 0}, {32, 64, 0}, {32, 64, 0}, {48, 96, 0}};
      const int np[12] = {2, 2, 2, 3, 3, 4, 2, 2, 3, 3, 3, 3}; const bool vt[12] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1}; const int pm[12] = {0, 0, 0, 0, 0, 0, 8, 8, 8, 8, 16, 8};
      for (int i = 0; i < 12; ++i) t.bud[i] += sim_budget(q, b[i], np[i], vt[i], pm[i], 80); }
    { const int wg[8] = {256, 512, 1024, 1024, 1024, 1024, 2048, 1024}, ev[8] = {8, 8, 8, 4, 16, 12, 8, 8}; const double rc[8] = {100, 100, 100, 100, 100, 100, 100, 0};
      for (int i = 0; i < 8; ++i) t.rep[i] += sim_repack(q, wg[i], ev[i], rc[i]); }
    { const int ks[3] = {2, 4, 8};
      for (int i = 0; i < 3; ++i) for (int w = 0; w < n; w += 64 * ks[i]) { int m = std::min(64 * ks[i], n - w); t.multi_s[i] += sim_multi(q.data() + w, m, ks[i], false); t.multi_d[i] += sim_multi(q.data() + w, m, ks[i], true); } }
    t.refill8 += sim_refill(q, 8); t.refill24 += sim_refill(q, 24);
    { const int pk[12] = {1, 1, 1, 2, 2, 2, 2, 2, 1, 2, 1, 3}, pm[12] = {16, 24, 32, 16, 24, 32, 32, 48, 8, 24, 1, 32}, wm[12] = {8, 12, 16, 8, 12, 16, 24, 24, 4, 12, 1, 16}, im[12] = {24, 24, 24, 24, 24, 24, 24, 24, 24, 8, 24, 24};
      for (int i = 0; i < 12; ++i) t.spec[i] += sim_spec(q, im[i], pk[i], pm[i], wm[i], 10); }
    { const int parks[4] = {8, 16, 32, 48}; for (int i = 0; i < 4; ++i) t.vote[i] += sim_vote(q, 8, parks[i]); }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::vector<unsigned> key(n);
    for (int i = 0; i < n; ++i) key[i] = rays[i].tmax < 0.f ? 0xffffffffu : sort_key(bvh, rays[i]);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return key[x] < key[y]; });
    std::vector<const Path *> qs(n);
    for (int i = 0; i < n; ++i) qs[i] = &paths[order[i]];
    for (int w = 0; w < n; w += 64) { int m = std::min(64, n - w); t.ww_sorted += sim_ww(qs.data() + w, m); }
    t.refill_sorted += sim_refill(qs, 8);
    if (std::getenv("RDR_TRACE_SIM_LAUNCHES")) {
        double u = 0, c0 = 0, c1 = 0; long live = 0;
        for (int i = 0; i < n; ++i) { u += useful_of(paths[i]); if (!(rays[i].tmax < 0.f)) ++live; }
        for (int w = 0; w < n; w += 64) { int m = std::min(64, n - w); c0 += sim_ifif(q.data() + w, m); c1 += sim_ww(qs.data() + w, m); }
        static int ordinal = 0;
        std::fprintf(stderr, "[trace_sim] launch %3d %s n=%7d live=%7ld useful/ray=%6.0f wave-cost/ray: plain %6.0f sorted %6.0f  eff %.2f -> %.2f\n", ordinal++, any ? "any    " : "closest", n, live, u / n, c0 * 64 / n, c1 * 64 / n, u / (c0 * 64), u / (c1 * 64));
    }
    { const int parks[4] = {8, 16, 32, 48}; for (int i = 0; i < 4; ++i) t.vote_sorted[i] += sim_vote(qs, 8, parks[i]); }
}

} // namespace tracesim
