"""HistorySplit's per-frame decision (tracking/HistorySplit.cpp:52-312): the CPU restatement against cases worked by hand from the
reference's text, and the host code (trex_amd/host/HipHistorySplit.h) against the restatement on random frames."""
import os
import subprocess
import numpy as np
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_individuals_one_blob():
    # both individuals are mapped to blob 0 and have no other edge: clique {0, 1} x {0}, 2 > 1 (:172).  Individual 0 (distance 1) keeps the
    # blob, individual 1 (distance 2) runs out of alternatives (:272-276): its closest blob is blob 0, which is assigned -> +1 for the
    # individual that holds it, +1 for itself (:285-298): expect[0] = 2, centers = [individual 0, individual 1]
    num, allow, big, centers = oracle.history_split(1, 2, {0: [0, 1]}, {0: [(0, 1.0)], 1: [(0, 2.0)]})
    assert list(num) == [2] and list(big) == [1] and list(allow) == [0] and centers == [[0, 1]]


def test_as_many_blobs_as_individuals_is_left_alone():
    # two individuals mapped to blob 0, the second one also paired with blob 1: the clique has 2 individuals and 2 blobs (:172) -> nothing
    num, allow, big, centers = oracle.history_split(2, 2, {0: [0, 1], 1: [1]}, {0: [(0, 1.0)], 1: [(0, 2.0), (1, 3.0)]})
    assert list(num) == [0, 0] and list(big) == [0, 0]


def test_three_individuals_two_blobs():
    # individuals 0, 1, 2; blobs 0, 1.  0 -> blob 0 (1.0); 1 -> blob 0 (0.5), blob 1 (4.0); 2 -> blob 0 (2.0).  Blob 0 has three individuals mapped.
    # Queue 0, 1, 2: 0 takes blob 0; 1 is closer (0.5 < 1.0): takes it over, 0 is queued again; 2: blob 0 held by 1 (0.5 <= 2.0): erased, queued
    # again, then empty; 0 again: blob 0 held by 1 (0.5 <= 1.0): erased, empty.  Individuals 0 and 2 have no alternative left; both are closest
    # to blob 0: the first adds the holder (1) and itself, the second itself: expect[0] = 3; individual 1 keeps a non-empty set (:272)
    num, allow, big, centers = oracle.history_split(2, 3, {0: [0, 1, 2], 1: [1]}, {0: [(0, 1.0)], 1: [(0, 0.5), (1, 4.0)], 2: [(0, 2.0)]})
    assert list(num) == [3, 0] and list(big) == [1, 0] and centers[0] == [1, 0, 2]


def test_displaced_individual_moves_to_its_second_blob():
    # 3 individuals, 2 blobs, all mapped to blob 0: 0 -> b0 (1.0), b1 (5.0); 1 -> b0 (0.5); 2 -> b0 (3.0).  1 displaces 0, which falls back to blob 1;
    # only individual 2 is left without an alternative: blob 0 (held by 1) is expected to hold 2
    num, allow, big, centers = oracle.history_split(2, 3, {0: [0, 1, 2]}, {0: [(0, 1.0), (1, 5.0)], 1: [(0, 0.5)], 2: [(0, 3.0)]})
    assert list(num) == [2, 0] and centers[0] == [1, 2]


def test_manual_split_and_switch():
    # manual_splits: expect = 2, never allow_less_than (:31-34), also with the history split switched off (:63-68); an unknown blob id is skipped (:29)
    num, allow, big, _ = oracle.history_split(2, 2, {0: [0, 1]}, {0: [(0, 1.0)], 1: [(0, 2.0)]}, manual=[1, 7], history_split_on=False)
    assert list(num) == [0, 2] and list(big) == [0, 1] and list(allow) == [0, 0]
    # a manually split blob is "already walked": the clique search does not start from it (:82), its expectation stays 2
    num, _, big, _ = oracle.history_split(1, 3, {0: [0, 1, 2]}, {0: [(0, 1.0)], 1: [(0, 2.0)], 2: [(0, 3.0)]}, manual=[0])
    assert list(num) == [2] and list(big) == [1]


def test_streak_threshold_drops_young_tracklets():
    # track_history_split_threshold = 5: individual 1's tracklet is 2 frames long -> it is not part of the clique (:104-148): 1 individual, 1 blob
    num, _, big, _ = oracle.history_split(1, 2, {0: [0, 1]}, {0: [(0, 1.0)], 1: [(0, 2.0)]}, streak=[9, 2], split_threshold=5)
    assert list(num) == [0] and list(big) == [0]
    num, _, big, _ = oracle.history_split(1, 2, {0: [0, 1]}, {0: [(0, 1.0)], 1: [(0, 2.0)]}, streak=[9, 5], split_threshold=5)
    assert list(num) == [2]


def test_host_code_matches_the_restatement_on_random_frames(tmp_path):
    oracle.build()
    exe = str(tmp_path / "test_history_split")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_history_split.cpp"),
                           "-o", exe, "-L", os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    out = subprocess.run([exe, "6000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "history split ok: 6000 cases" in out.stdout, out.stdout + out.stderr
