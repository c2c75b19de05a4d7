// moment_schedule.h -- host-side schedule model of the gradient's LDS-resident moment pass (plain C++: no device code, so that the
// CPU test suite can compile and check it with g++: tests/test_host_logic.py).
#pragma once
#include <cstddef>
#include <vector>

namespace gpmpc_hip {

// Schedule model of the LDS-resident moment pass (pair_moments_kernel).  Its work items are (row chunk of CH rows) x (64 column
// units); the NW wavefronts of a workgroup pull them from a queue, an item costs its longest lane's rows plus a fixed part (column
// set-up, the fold of the accumulators: ~12 rows' worth of instructions), and the wavefronts of a SIMD share its issue slots.
// With 64-row chunks config 2 has 21 items for 16 wavefronts -- one and a third rounds, the last one on 5 wavefronts -- and config 1
// (N = 50) ONE item per pair: 3 of 16 wavefronts busy.  The model plays the queue for a chunk length and returns the time of the
// busiest workgroup in row units; launch_rollout_grad takes the shortest (ties: the longer chunk).
struct MomentItems { std::vector<int> diag, full; };          // rows of every item of a diagonal / an off-diagonal pair (0: no valid lane)

inline MomentItems moment_items(int N, int NC, int CH) {
    const int NCU = (N + NC - 1) / NC, RC = (N + CH - 1) / CH, wpp = (RC * NCU + 63) / 64;
    std::vector<int> tri(RC + 1);
    int run = 0;
    for (int r = 0; r <= RC; ++r) { tri[r] = run; const int first = (r * CH) / NC; run += first < NCU ? NCU - first : 0; }
    MomentItems it;
    it.diag.assign(wpp, 0); it.full.assign(wpp, 0);
    for (int r = 0; r < RC; ++r) {
        const int i0 = r * CH, i1 = (i0 + CH < N) ? i0 + CH : N;
        for (int jc = 0; jc < NCU; ++jc) {
            const int flat = r * NCU + jc;
            if (it.full[flat / 64] < i1 - i0) it.full[flat / 64] = i1 - i0;
        }
        const int first = (r * CH) / NC;
        for (int jc = first; jc < NCU; ++jc) {
            const int flat = tri[r] + (jc - first);
            int jl = NC * jc + NC - 1; if (jl > N - 1) jl = N - 1;
            int e1 = i1; if (e1 > jl + 1) e1 = jl + 1;
            const int len = e1 > i0 ? e1 - i0 : 0;
            if (it.diag[flat / 64] < len) it.diag[flat / 64] = len;
        }
    }
    for (int& v : it.diag) v = (v + 3) & ~3;
    for (int& v : it.full) v = (v + 3) & ~3;
    return it;
}

inline double moment_schedule_cost(const MomentItems& it, const std::vector<int>& pair_is_diag, int G, int gz, int NW) {
    constexpr double kItemFixed = 12.0, kEmptyItem = 0.5, kGroupFixed = 8.0, kOneWaveRate = 0.4;
    std::vector<double> per_z(gz > 0 ? gz : 1, 0.0);
    int gi = 0;
    for (std::size_t q0 = 0; q0 < pair_is_diag.size(); q0 += G, ++gi) {
        std::vector<double> queue;
        for (std::size_t q = q0; q < q0 + G && q < pair_is_diag.size(); ++q)
            for (int rows : (pair_is_diag[q] ? it.diag : it.full)) queue.push_back(rows > 0 ? rows + kItemFixed : kEmptyItem);
        std::vector<double> left(NW, 0.0);
        std::size_t next = 0;
        for (int w = 0; w < NW && next < queue.size(); ++w) left[w] = queue[next++];
        double t = 0.0;
        for (;;) {
            double rate[64];
            double dt = -1.0;
            for (int sd = 0; sd < 4; ++sd) {
                int n = 0;
                for (int w = sd; w < NW; w += 4) n += left[w] > 0.0;
                const double share = n ? ((n * kOneWaveRate < 1.0 ? n * kOneWaveRate : 1.0) / n) : 0.0;
                for (int w = sd; w < NW; w += 4) rate[w] = share;
            }
            for (int w = 0; w < NW; ++w)
                if (left[w] > 0.0 && (dt < 0.0 || left[w] / rate[w] < dt)) dt = left[w] / rate[w];
            if (dt < 0.0) break;
            t += dt;
            for (int w = 0; w < NW; ++w)
                if (left[w] > 0.0) {
                    left[w] -= rate[w] * dt;
                    if (left[w] < 1e-9) left[w] = next < queue.size() ? queue[next++] : 0.0;
                }
        }
        per_z[gi % per_z.size()] += t + kGroupFixed;
    }
    double worst = 0.0;
    for (double v : per_z) if (v > worst) worst = v;
    return worst;
}


// The chunk length (multiple of 4, 8 <= CH <= CH0) with the shortest modelled pass; `pairs_per_group(CH)` returns how many pairs'
// row records fit the LDS together at that chunk length (0: the layout does not fit) and `spread` the number of workgroups the
// groups of one (candidate, step) are spread over (small batches).  Ties go to the longer chunk.
template <class GroupsOf>
inline int choose_moment_chunk(int N, int NC, int NW, int CH0, const std::vector<int>& pair_is_diag, GroupsOf&& groups_of) {
    int want = CH0;
    double best = 0.0;
    for (int c = CH0; c >= 8; c -= 4) {
        int G = 0, gz = 1;
        if (!groups_of(c, G, gz) || G <= 0) continue;
        const double cost = moment_schedule_cost(moment_items(N, NC, c), pair_is_diag, G, gz, NW);
        if (best == 0.0 || cost < best * 0.995) { best = cost; want = c; }
    }
    return want;
}

}  // namespace gpmpc_hip
