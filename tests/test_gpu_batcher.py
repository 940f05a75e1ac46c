"""the library's own batch accumulator + dispatcher (mm2gb_batcher_*): reads fed one at a time, from several
threads, over several engines; every read's chains equal the oracle's."""
import threading

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

mm = pytest.importorskip("mm2gb_amd")


def reads_and_oracle(seed, n_reads, lo, hi):
    a, off = mm.synth_reads(seed, 0, n_reads, lo, hi)
    prm = orc.default_param()
    want = [orc.lchain_dp(a[off[r]:off[r + 1]], prm, want_fp=False) for r in range(n_reads)]
    return a, off, want


def small_config(max_total_n, max_read, min_n):
    c = mm.default_config()
    c.max_total_n, c.max_read, c.min_n = max_total_n, max_read, min_n
    c.has_max_total_n = c.has_max_read = 1
    c.score_kernel.micro_batch = 1
    return c


@pytest.mark.parametrize("post_threads", [3, 0], ids=["host-post-pass", "device-post-pass"])
def test_reads_in_batches_out_match_the_oracle(post_threads):
    a, off, want = reads_and_oracle(31, 40, 2_000, 40_000)
    tiny = [np.zeros((0, 2), np.uint64), a[off[3]:off[3] + 7], a[off[5]:off[5] + 60]]
    with mm.Batcher(devices=[0, 0], config=small_config(60_000, 8, 100), post_threads=post_threads) as b:
        for r in range(40):
            b.add(r, a[off[r]:off[r + 1]])
            if r % 13 == 0:
                for k, t in enumerate(tiny):
                    b.add(1000 + 10 * r + k, t)
        b.flush()
        st = b.stats()
        assert st["reads"] == 40 + 4 * 3 and st["reads_per_lane"][1] == 12 and st["batches"][0] >= 5 and st["n_engines"] == 2
        assert sum(st["batches_per_engine"]) == sum(st["batches"])
        for r in range(40):
            u, ao = b.results[r]
            assert np.array_equal(u, want[r]["u"]) and np.array_equal(ao, want[r]["a_out"]), f"read {r}"
        prm = orc.default_param()
        for r in range(0, 40, 13):
            for k, t in enumerate(tiny):
                o = orc.lchain_dp(t, prm, want_fp=False)
                u, ao = b.results[1000 + 10 * r + k]
                assert np.array_equal(u, o["u"]) and np.array_equal(ao, o["a_out"])
        # a second round on the same batcher after a flush
        b.results.clear()
        for r in range(5):
            b.add(r, a[off[r]:off[r + 1]])
        b.flush()
        for r in range(5):
            assert np.array_equal(b.results[r][0], want[r]["u"])


def test_several_producer_threads_and_a_read_larger_than_the_limit():
    a, off, want = reads_and_oracle(77, 24, 5_000, 60_000)
    with mm.Batcher(devices=[0, 0, 0], config=small_config(500, 4, 0), post_threads=2) as b:       # every read is over the anchor limit
        def feed(lo, hi):
            for r in range(lo, hi):
                b.add(r, a[off[r]:off[r + 1]])
        th = [threading.Thread(target=feed, args=(k * 8, k * 8 + 8)) for k in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        b.flush()
        assert b.stats()["batches"][0] == 24
        for r in range(24):
            assert np.array_equal(b.results[r][0], want[r]["u"]) and np.array_equal(b.results[r][1], want[r]["a_out"]), f"read {r}"


@pytest.mark.parametrize("post_threads", [2, 0], ids=["host-post-pass", "device-post-pass"])
def test_native_producers_copy_side_by_side(post_threads):
    """mm2gb_batcher_feed: six producer threads of the library add the reads of a packed batch one at a time; a read's place in the open
    batch is reserved under the batcher's lock and the read is copied OUTSIDE it, batches close under the producers'
    feet and their buffers grow while others copy.  Every read's chains equal the oracle's, twice over on the same batcher."""
    a, off, want = reads_and_oracle(91, 60, 2_000, 30_000)
    with mm.Batcher(devices=[0, 0], config=small_config(90_000, 7, 0), post_threads=post_threads) as b:
        for rnd in range(2):
            b.results.clear()
            b.feed(1000 * rnd, a, off, producers=6)
            b.flush()
            assert len(b.results) == 60
            for r in range(60):
                u, ao = b.results[1000 * rnd + r]
                assert np.array_equal(u, want[r]["u"]) and np.array_equal(ao, want[r]["a_out"]), f"round {rnd} read {r}"
        assert b.stats()["reads"] == 120
