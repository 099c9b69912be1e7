"""CPU-side checks of the product boundary: libagx.so loads, exports every symbol include/agx.h
declares, refuses to compute without a device (no fallback), and its text readers see exactly
the pairs the reference programs see (checked by scoring them with the oracle against the goldens)."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

import accelerating_genomics_amd.api as agx
import accelerating_genomics_amd.synth as synth
from tests.test_oracle_sw import expect_scores

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(agx.LIB_PATH):
        agx.build()


def test_header_and_library_agree_on_symbols():
    hdr = open(os.path.join(ROOT, "include", "agx.h")).read()
    declared = set(re.findall(r"\b(agx_[A-Za-z0-9_]+)\s*\(", hdr))
    assert declared == set(agx.SYMBOLS)
    lib = ctypes.CDLL(agx.LIB_PATH)
    for s in agx.SYMBOLS:
        assert hasattr(lib, s), s


def test_no_device_means_loud_failure_not_fallback():
    if agx.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(agx.AgxError) as e:
        agx.Context(0)
    assert e.value.code == agx.E_NODEVICE
    b = synth.sw_pairs(2, 5, 9, seed=1)
    with pytest.raises(agx.AgxError) as e:
        agx.sw_score_multi(b)
    assert e.value.code == agx.E_NODEVICE
    with pytest.raises(agx.AgxError):
        agx.phmm_forward_multi(synth.phmm_regions(1, 1, 1, 5, 9, seed=1))


def test_product_never_touches_the_oracle():
    for path in glob.glob(os.path.join(ROOT, "accelerating-genomics_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".c", ".cpp", ".h", ".hip", "Makefile")):
            txt = open(path, errors="replace").read()
            assert "liboracle" not in txt and "oracle_api" not in txt and "oracle/" not in txt, path


SW_CASES = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "sw_*.in")))


@pytest.mark.parametrize("name", SW_CASES)
def test_sw_text_reader_sees_the_reference_pairs(oracle, golden_dir, name):
    n_ref, s_ref = expect_scores(os.path.join(golden_dir, name + ".expect"))
    line_num, b, dangling = agx.read_sw_text(os.path.join(golden_dir, name + ".in"))
    assert line_num == n_ref and b.n_pairs == s_ref.size
    assert np.array_equal(oracle.sw_batch(b), s_ref)
    if name == "sw_oddlines":
        assert dangling is not None and open(os.path.join(golden_dir, name + ".expect"), "rb").read().endswith(dangling)
    else:
        assert dangling is None


def test_sw_text_reader_splits_long_lines_like_fgets(tmp_path, oracle):
    p = tmp_path / "long.in"
    p.write_bytes(b"2\n" + b"A" * 1500 + b"\n" + b"C" * 10 + b"\n")
    _, b, _ = agx.read_sw_text(str(p))  # 1000-byte buffer: 999 + 501+'\n' become the pair, 'C...' is never read
    assert b.n_pairs == 1 and list(b.len) == [999, 502]
    _, b, _ = agx.read_sw_text(str(p), 10000)  # hipvers.cpp:40 buffer
    assert list(b.len) == [1501, 11]


def test_sw_text_reader_errors(tmp_path):
    with pytest.raises(agx.AgxError) as e:
        agx.read_sw_text(str(tmp_path / "missing"))
    assert e.value.code == agx.E_IO
    (tmp_path / "empty").write_bytes(b"")
    with pytest.raises(agx.AgxError) as e:
        agx.read_sw_text(str(tmp_path / "empty"))
    assert "file is empty" in str(e.value)


@pytest.mark.parametrize("name", ["phmm_test", "phmm_10s", "phmm_synth", "phmm_far", "phmm_long"])
def test_phmm_text_reader_matches_python_mirror(golden_dir, name):
    path = os.path.join(golden_dir, name + ".in")
    got, seen, trunc = agx.read_phmm_text(path)
    want = synth.parse_phmm_text(open(path, "rb").read())
    assert trunc == 0 and seen == want.n_regions
    for f in ("read_bases", "q_base", "q_ins", "q_del", "q_gcp", "roff", "hap_bases", "hoff", "rreg", "hreg"):
        assert np.array_equal(getattr(got, f), getattr(want, f)), f


def test_phmm_text_reader_truncated_region(tmp_path, golden_dir):
    data = open(os.path.join(golden_dir, "phmm_synth.in"), "rb").read().split(b"\n")
    (tmp_path / "cut.in").write_bytes(b"\n".join(data[:14]) + b"\n")  # region 1 = 1+6+4 lines, region 2 cut short
    got, seen, trunc = agx.read_phmm_text(str(tmp_path / "cut.in"))
    assert trunc == 1 and seen == 2 and got.n_regions == 1 and got.n_pairs == 24


def _fgets_lines(data: bytes, bufsz: int):
    """What a loop of fgets(buf, bufsz) + strlen(buf) sees (antidiagonalSmithWaterman.c:219-247)."""
    pos = 0
    while pos < len(data):
        chunk = data[pos:pos + bufsz - 1]
        i = chunk.find(b"\n")
        line = chunk[:i + 1] if i >= 0 else chunk
        pos += len(line)
        z = line.find(b"\0")
        yield line if z < 0 else line[:z]


@pytest.mark.parametrize("seed,final_newline", [(1, True), (2, False), (3, True)])
def test_block_reader_has_fgets_semantics_and_chunks_concatenate(tmp_path, seed, final_newline):
    """The memchr block reader against a byte-level model of fgets/strlen: over-long lines split at
    bufsz-1, a NUL hides the rest of its line, the last line may lack its newline; reading in chunks
    (agx_sw_reader_next) gives the same pairs as reading at once."""
    rng = np.random.default_rng(seed)
    lens = [0, 1, 5, 998, 999, 1000, 1001, 1997, 1998, 1999, 2500] + [int(x) for x in rng.integers(1, 1200, size=90)]
    rng.shuffle(lens)
    body = b""
    for k, n in enumerate(lens):
        line = bytearray(rng.choice(list(b"ACGT"), size=n).tolist())
        if k % 17 == 5 and n > 3:
            line[n // 2] = 0
        body += bytes(line) + b"\n"
    if not final_newline:
        body = body[:-1]
    for header in (b"100000\n", b"61\n", b"6\n"):
        p = tmp_path / "f.in"
        p.write_bytes(header + body)
        want = list(_fgets_lines(body, 1000))[: int(header) + (int(header) & 1)]
        dangling = want[-1] if len(want) % 2 else None
        want = want[: len(want) // 2 * 2]
        line_num, b, dang = agx.read_sw_text(str(p))
        assert line_num == int(header) and b.n_pairs == len(want) // 2
        assert [b.seq(k) for k in range(2 * b.n_pairs)] == want
        assert dang == dangling
        for chunk in (1, 7, 10 ** 6):
            got, dangs = [], []
            for ln, cb, d in agx.read_sw_text_chunks(str(p), chunk):
                assert ln == int(header) and cb.n_pairs <= chunk
                got += [cb.seq(k) for k in range(2 * cb.n_pairs)]
                dangs.append(d)
            assert got == want and [d for d in dangs if d is not None] == ([dangling] if dangling is not None else [])


@pytest.mark.parametrize("final_newline,odd", [(True, False), (False, True)])
def test_parallel_chunk_reader_against_the_fgets_model(tmp_path, final_newline, odd):
    """Chunks of 8 MB and more of a regular file are read and scanned by the host thread pool, every thread owning the
    physical lines that start in its slice (agx_text.c).  A 30 MB file with over-long lines, NUL bytes, an optional
    missing final newline and an odd line count, read at once and in chunks, against the byte-level model."""
    rng = np.random.default_rng(99)
    lens = rng.integers(0, 700, size=90000)
    lens[rng.integers(0, lens.size, size=300)] = rng.integers(990, 4200, size=300)  # lines fgets splits
    if odd:
        lens = lens[:-1]
    data = rng.integers(65, 91, size=int(lens.sum()) + lens.size, dtype=np.uint8)
    ends = np.cumsum(lens + 1) - 1
    data[ends] = 10
    nul = rng.integers(0, data.size, size=200)
    data[nul[data[nul] != 10]] = 0
    body = data.tobytes() if final_newline else data.tobytes()[:-1]
    assert len(body) > 24 << 20
    lines = list(_fgets_lines(body, 1000))
    p = tmp_path / "big.in"
    for count in (10 ** 9, len(lines) - 5):
        p.write_bytes(b"%d\n" % count + body)
        want = lines[: count + (count & 1)]
        dangling = want[-1] if len(want) % 2 else None
        want = want[: len(want) // 2 * 2]
        line_num, b, dang = agx.read_sw_text(str(p))
        assert line_num == count and b.n_pairs == len(want) // 2 and dang == dangling
        off, ln = b.off.astype(np.int64), b.len.astype(np.int64)
        raw = b.bases.tobytes()
        assert all(raw[off[k] : off[k] + ln[k]] == want[k] for k in range(0, 2 * b.n_pairs, 37))
        assert [raw[off[k] : off[k] + ln[k]] for k in range(2 * b.n_pairs - 50, 2 * b.n_pairs)] == want[-50:]
        want_len = np.array([len(x) for x in want], dtype=np.int64)
        assert np.array_equal(ln, want_len)
        for chunk in (20000, 10 ** 6):
            n_got, dangs, k0 = 0, [], 0
            for lnum, cb, d in agx.read_sw_text_chunks(str(p), chunk):
                assert lnum == count and cb.n_pairs <= chunk
                cl = cb.len.astype(np.int64)
                assert np.array_equal(cl, want_len[k0 : k0 + cl.size])
                craw, coff = cb.bases.tobytes(), cb.off.astype(np.int64)
                for k in list(range(0, cl.size, 101)) + list(range(max(0, cl.size - 4), cl.size)):
                    assert craw[coff[k] : coff[k] + cl[k]] == want[k0 + k]
                k0 += cl.size
                dangs.append(d)
            assert k0 == len(want) and [d for d in dangs if d is not None] == ([dangling] if dangling is not None else [])


@pytest.mark.parametrize("name", ["phmm_10s", "phmm_synth", "phmm_long"])
@pytest.mark.parametrize("max_pairs", [1, 100, 10 ** 9])
def test_phmm_reader_hands_out_whole_regions(golden_dir, name, max_pairs):
    """agx_phmm_reader_*: the reference's batch loop (antidiagsPairHMM.c:371-433,484-489) in pieces.  Chunks are whole
    regions, at least one per call, cut at the first region that reaches max_pairs; together they are the file."""
    path = os.path.join(golden_dir, name + ".in")
    whole, seen, trunc = agx.read_phmm_text(path)
    chunks = list(agx.read_phmm_text_chunks(path, max_pairs))
    assert sum(c[1] for c in chunks) == seen and not any(c[2] for c in chunks) and trunc == 0
    assert sum(c[0].n_regions for c in chunks) == whole.n_regions and sum(c[0].n_pairs for c in chunks) == whole.n_pairs
    for c, _, _ in chunks[:-1]:
        assert c.n_regions >= 1
        if max_pairs < 10 ** 9:
            before_last = c.n_pairs - int((c.rreg[-1] - c.rreg[-2]) * (c.hreg[-1] - c.hreg[-2]))
            assert before_last < max_pairs <= c.n_pairs or c.n_regions == 1
    for f in ("read_bases", "q_base", "q_ins", "q_del", "q_gcp", "hap_bases"):
        assert np.array_equal(np.concatenate([getattr(c[0], f) for c in chunks]), getattr(whole, f)), f
    assert np.array_equal(np.concatenate([np.diff(c[0].roff) for c in chunks]), np.diff(whole.roff))
    assert np.array_equal(np.concatenate([np.diff(c[0].hoff) for c in chunks]), np.diff(whole.hoff))


def test_phmm_reader_truncated_region_ends_the_stream(tmp_path, golden_dir):
    data = open(os.path.join(golden_dir, "phmm_synth.in"), "rb").read().split(b"\n")
    (tmp_path / "cut.in").write_bytes(b"\n".join(data[:14]) + b"\n")  # region 1 complete, region 2 cut short
    chunks = list(agx.read_phmm_text_chunks(str(tmp_path / "cut.in"), 1))
    assert [(c[0].n_regions, c[1], c[2]) for c in chunks] == [(1, 1, 0), (0, 1, 1)]


def test_shard_cut_rules_match_the_python_mirror():
    """agx_sw_shard_cuts / agx_phmm_shard_cuts (what agx_*_devices and bench.py's strong-scaling leg cut by) against
    dist.shard_bounds, the rule bench.py's ranks use: contiguous, balanced by cells, every unit in exactly one shard."""
    import accelerating_genomics_amd.dist as agd

    for seed, n in ((1, 1), (2, 7), (3, 1000), (4, 20000)):
        b = synth.sw_pairs(n, 1, 400, seed=seed)
        cost = b.len[0::2].astype(np.float64) * b.len[1::2].astype(np.float64)
        for shards in (1, 2, 3, 8, 64):
            cut = agx.sw_shard_cuts(b, shards)
            assert np.array_equal(cut, agd.shard_bounds(cost, shards)), (n, shards)
            assert cut[0] == 0 and cut[-1] == n and np.all(np.diff(cut) >= 0)
            if n >= 1000 and shards <= 8:
                per = np.add.reduceat(cost, cut[:-1])
                assert per.max() / per.mean() < 1.05
    p = synth.phmm_regions(37, 5, 3, 80, 160, seed=5, jitter=40)
    rb = np.diff(p.roff[p.rreg].astype(np.float64))
    hb = np.diff(p.hoff[p.hreg].astype(np.float64))
    for shards in (1, 2, 5, 8, 50):
        cut = agx.phmm_shard_cuts(p, shards)
        assert np.array_equal(cut.astype(np.int64), agd.shard_bounds(rb * hb, shards))
    assert list(agx.sw_shard_cuts(synth.sw_from_seqs([]), 4)) == [0, 0, 0, 0, 0]
