// HipHistorySplit::decide (trex_amd/host/HipHistorySplit.h) against the C restatement of HistorySplit's decision (oracle/trex_split.c) on random
// frames: cliques of individuals and blobs, distance ties, manual splits, the streak threshold.  CPU only (no device call is made).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../trex_amd/host/HipHistorySplit.h"
#include "../../oracle/trex_oracle.h"

using namespace track;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d, case %d)\n", #c, __LINE__, it); return 1; } } while (0)

int main(int argc, char** argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 4000;
    std::mt19937 rng(11);
    auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };
    int n_split_cases = 0, n_big_total = 0, n_three = 0;
    for (int it = 0; it < cases; ++it) {
        const int nb = 1 + rnd(8), nf = 1 + rnd(10);
        HipHistorySplit::Frame F;
        HipHistorySplit::Settings S;
        S.track_do_history_split = rnd(12) != 0;
        S.track_history_split_threshold = rnd(3) == 0 ? rnd(6) : -1;
        std::vector<int32_t> map_off(nb + 1, 0), map_fish, pair_off(nf + 1, 0), pair_blob, streak(nf), manual;
        std::vector<float> pair_d;
        std::vector<std::vector<int>> mapped(nb);
        const int dense = 1 + rnd(4), tie = rnd(2);
        for (int f = 0; f < nf; ++f) {
            streak[f] = rnd(8) - 1;
            F.valid_frame_streak[f] = streak[f];
            F.last_positions[f] = {cmn::Vec2((float)f, 1.f)};
            pair_off[f] = (int32_t)pair_blob.size();
            std::vector<std::pair<HipHistorySplit::bid_t, float>> edges;
            for (int b = 0; b < nb; ++b)
                if (rnd(4) < dense && !(rnd(5) == 0)) {
                    const float d = tie ? (float)(1 + rnd(3)) : (float)(1 + rnd(1000)) * 0.37f;
                    edges.emplace_back((HipHistorySplit::bid_t)b, d);
                    pair_blob.push_back(b); pair_d.push_back(d);
                    if (rnd(6) != 0) mapped[b].push_back(f);           // blob_mappings: the individuals whose edge also passed the distance test
                }
            if (!edges.empty() || rnd(2)) F.paired[f] = edges;
        }
        pair_off[nf] = (int32_t)pair_blob.size();
        for (int b = 0; b < nb; ++b) {
            F.blob_pos[(HipHistorySplit::bid_t)b] = cmn::Vec2(0.f, 0.f);
            if (rnd(9) == 0) mapped[b].push_back(-1);                    // an invalid Idx_t (a manual split's entry)
            map_off[b] = (int32_t)map_fish.size();
            std::set<int32_t> st(mapped[b].begin(), mapped[b].end());
            for (int32_t f : st) map_fish.push_back(f);
            if (!st.empty() || rnd(2)) F.blob_mappings[(HipHistorySplit::bid_t)b] = std::set<HipHistorySplit::Idx_t>(st.begin(), st.end());
        }
        map_off[nb] = (int32_t)map_fish.size();
        if (rnd(5) == 0) { const int m = rnd(nb + 2); manual.push_back(m < nb ? m : -1); if (m < nb) F.manual_splits.push_back((HipHistorySplit::bid_t)m); else F.manual_splits.push_back(9999u); }

        const HipHistorySplit::Decision D = HipHistorySplit::decide(F, S);
        std::vector<int32_t> number(nb), coff(nb + 1), cfish(2 * nf + 2);
        std::vector<uint8_t> allow(nb), big(nb);
        const int n_big = oracle_history_split(nb, nf, map_off.data(), map_fish.data(), pair_off.data(), pair_blob.data(), pair_d.data(), streak.data(),
                                               S.track_history_split_threshold, manual.data(), (int32_t)manual.size(), S.track_do_history_split ? 1 : 0,
                                               number.data(), allow.data(), big.data(), coff.data(), cfish.data());
        CHECK(n_big == (int)D.big_blobs.size());
        for (int b = 0; b < nb; ++b) {
            auto x = D.expect.find((HipHistorySplit::bid_t)b);
            const size_t num = x == D.expect.end() ? 0 : x->second.number;
            CHECK((int)num == number[b]);
            CHECK((x == D.expect.end() ? false : x->second.allow_less_than) == (allow[b] != 0));
            bool in_big = false;
            for (auto q : D.big_blobs) in_big |= q == (HipHistorySplit::bid_t)b;
            CHECK(in_big == (big[b] != 0));
            const size_t nc = x == D.expect.end() ? 0 : x->second.centers.size();
            CHECK((int)nc == coff[b + 1] - coff[b]);
            for (size_t k = 0; k < nc; ++k) {
                CHECK(x->second.centers[k].size() == 1);
                CHECK((int)x->second.centers[k][0].x == cfish[coff[b] + (int)k]);       // last_positions[f] = (f, 1), bounds().pos() = (0, 0)
            }
            if (number[b] >= 3) ++n_three;
        }
        n_big_total += n_big;
        n_split_cases += n_big > 0;
    }
    std::printf("history split ok: %d cases, %d with a split, %d big blobs, %d expectations of three or more\n", cases, n_split_cases, n_big_total, n_three);
    return 0;
}
