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


def _phmm_reference_reading_model(data: bytes):
    """The reference's reading of a PairHMM file byte for byte (antidiagsPairHMM.c:353-433): fgets into a 5001-byte
    buffer, strcspn(line, "\n") / C-string ends, a header's sscanf("%d %d") leaving nr / nh untouched where it fails,
    the region cut short by the end of the file, (strlen - 4) / 5 bases per read line and its five sscanf("%s") fields.
    -> ("ok", regions, headers seen, truncated) or ("err", region number, read index, kind)."""
    import re

    pos = 0

    def fgets():
        nonlocal pos
        if pos >= len(data):
            return None
        piece = data[pos:pos + 5000]
        k = piece.find(b"\n")
        if k >= 0:
            piece = piece[:k + 1]
        pos += len(piece)
        return piece

    def text(line):
        k = line.find(b"\n")
        line = line if k < 0 else line[:k]
        z = line.find(b"\0")
        return line if z < 0 else line[:z]

    nr = nh = 0
    regions, seen, truncated = [], 0, 0
    while True:
        head = fgets()
        if head is None:
            break
        seen += 1
        m = re.match(rb"[ \t\n\v\f\r]*([+-]?\d+)(?:[ \t\n\v\f\r]*([+-]?\d+))?", text(head))
        if m:
            nr = int(m.group(1))
            if m.group(2) is not None:
                nh = int(m.group(2))
        if nh < 0:          # :381-385: malloc of a negative count fails, "Memory allocation failed for haplotypes array"
            truncated = 2
            break
        nr = max(nr, 0)     # a negative read count runs none of the loops over reads
        reads, haps = [], []
        while len(reads) < nr:
            line = fgets()
            if line is None:
                break
            reads.append(text(line))
        if len(reads) == nr:
            while len(haps) < nh:
                line = fgets()
                if line is None:
                    break
                haps.append(text(line))
        if len(reads) < nr or len(haps) < nh:
            truncated = 1
            break
        tracks = []
        for i, line in enumerate(reads):
            if len(line) < 4:
                return ("err", len(regions) + 1, i, "line too short")
            n = (len(line) - 4) // 5
            tok = re.findall(rb"[^ \t\n\v\f\r]+", line)[:5]
            tok += [b""] * (5 - len(tok))
            for k in range(5):
                if len(tok[k]) < n:
                    return ("err", len(regions) + 1, i, "field %d shorter" % k)
            tracks.append(tuple(t[:n] for t in tok))
        regions.append((tracks, haps))
    return ("ok", regions, seen, truncated)


def _nasty_phmm_files():
    rng = np.random.default_rng(7)
    seq = lambda n, alpha=b"ACGTN": bytes(rng.choice(np.frombuffer(alpha, np.uint8), n))
    q = lambda n: bytes(rng.integers(34, 80, size=n).astype(np.uint8))
    readline = lambda n, sep=b" ": sep.join([seq(n), q(n), q(n), q(n), q(n)])

    def region(nr, nh, R, H, sep=b" ", nl=b"\n"):
        out = [b"%d %d" % (nr, nh)]
        out += [readline(int(rng.integers(1, R + 1)), sep) for _ in range(nr)]
        out += [seq(int(rng.integers(1, H + 1)), b"ACGT") for _ in range(nh)]
        return nl.join(out) + nl

    f = {}
    f["plain"] = b"".join(region(int(rng.integers(1, 6)), int(rng.integers(1, 5)), 60, 90) for _ in range(40))
    f["nofinalnl"] = f["plain"][:-1]
    f["crlf"] = b"".join(region(2, 2, 30, 40, nl=b"\r\n") for _ in range(5))
    f["tabs"] = b"".join(region(3, 2, 30, 40, sep=b" \t ") for _ in range(5))
    f["longhap"] = b"2 2\n" + readline(50) + b"\n" + readline(70) + b"\n" + seq(6000, b"ACGT") + b"\n" + seq(12001, b"ACGT") + b"\n" + region(2, 2, 20, 20)
    f["longread"] = (b"1 1\n" + readline(999) + b"\n" + seq(100, b"ACGT") + b"\n1 1\n" + readline(1000) + b"\n" + seq(100, b"ACGT") +
                     b"\n1 1\n" + readline(1200) + b"\n" + seq(50, b"ACGT") + b"\n" + region(1, 1, 10, 10))
    f["exact5000"] = b"1 2\n" + readline(30) + b"\n" + seq(5000, b"ACGT") + b"\n" + seq(4999, b"ACGT") + b"\n" + region(1, 1, 10, 10)
    f["nul"] = (b"2 2\n" + readline(20) + b"\n" + readline(25)[:40] + b"\0" + readline(25)[40:] + b"\n" + seq(30, b"ACGT") + b"\nACG\0TTT\n" +
                region(1, 2, 10, 10))
    f["trunc_reads"] = region(2, 2, 20, 30) + b"3 2\n" + readline(10) + b"\n"
    f["trunc_haps"] = region(2, 2, 20, 30) + b"2 3\n" + readline(10) + b"\n" + readline(12) + b"\n" + seq(20, b"ACGT") + b"\n"
    f["badheader"] = (region(2, 2, 20, 30) + b"x y\n" + readline(10) + b"\n" + readline(12) + b"\n" + seq(20, b"ACGT") + b"\n" + seq(22, b"ACGT") +
                      b"\n" + region(1, 1, 10, 10))
    f["half_header"] = region(2, 3, 20, 30) + b"1\n" + readline(9) + b"\n" + seq(5, b"ACGT") + b"\n" + seq(6, b"ACGT") + b"\n" + seq(7, b"ACGT") + b"\n"
    f["emptyline_header"] = region(1, 1, 20, 30) + b"\n" + readline(10) + b"\n" + seq(20, b"ACGT") + b"\n"
    f["zero_counts"] = b"0 2\nACGT\nACG\n" + region(1, 1, 10, 10) + b"2 0\n" + readline(5) + b"\n" + readline(6) + b"\n" + region(1, 1, 10, 10)
    f["negative"] = b"-1 2\nACGT\nACG\n" + region(1, 1, 10, 10)
    f["negative_haps"] = region(2, 2, 10, 10) + b"2 -1\n" + region(1, 1, 10, 10)
    f["shortline"] = region(1, 1, 10, 10) + b"1 1\nAB\nACGT\n"
    f["shortfield"] = region(1, 1, 10, 10) + b"1 1\nACGTACGT II II II II\nACGT\n"
    f["empty"] = b""
    f["onlyheader"] = b"3 3\n"
    f["extra_fields"] = b"1 1\n" + readline(10) + b" extra stuff\n" + seq(10, b"ACGT") + b"\n"
    f["spaces_hap"] = b"1 2\n" + readline(10) + b"\nACGT ACGT\n  ACG\n"
    f["many"] = b"".join(region(12, 3, 120, 200) for _ in range(120))  # enough read lines for several threads
    return f


@pytest.mark.parametrize("max_pairs", [1, 7, 1 << 40])
def test_phmm_reader_against_the_byte_level_model(tmp_path, max_pairs):
    """agx_phmm_reader_* (block buffer with fgets' rule, read lines cut into their fields by the thread pool) against a
    byte-level model of the reference's reading, on files with every irregularity the format allows: lines beyond the
    5000-byte buffer, NUL bytes, CR LF, tabs, no final newline, malformed and half headers, zero and negative counts,
    regions cut short, short lines and fields (errors name the region and the read the reference would trip on)."""
    for name, data in _nasty_phmm_files().items():
        path = str(tmp_path / (name + ".in"))
        open(path, "wb").write(data)
        want = _phmm_reference_reading_model(data)
        try:
            chunks = list(agx.read_phmm_text_chunks(path, max_pairs))
        except agx.AgxError as e:
            assert want[0] == "err" and e.code == agx.E_IO, (name, want[:1], str(e))
            assert "region %d, read %d:" % (want[1], want[2]) in str(e) and want[3] in str(e), (name, want, str(e))
            continue
        assert want[0] == "ok", (name, want)
        _, regions, seen, truncated = want
        assert sum(c[1] for c in chunks) == seen and max([c[2] for c in chunks] + [0]) == truncated, name
        got = []
        for b, _, _ in chunks:
            for g in range(b.n_regions):
                reads = [tuple(x[int(b.roff[r]):int(b.roff[r + 1])].tobytes() for x in (b.read_bases, b.q_base, b.q_ins, b.q_del, b.q_gcp))
                         for r in range(int(b.rreg[g]), int(b.rreg[g + 1]))]
                haps = [b.hap_bases[int(b.hoff[h]):int(b.hoff[h + 1])].tobytes() for h in range(int(b.hreg[g]), int(b.hreg[g + 1]))]
                got.append((reads, haps))
        assert got == regions, name


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


def test_one_shot_sw_read_of_a_file_beyond_64_mb(tmp_path):
    """agx_sw_text_read made ONE reader call whose threaded branch read at most its 64 MB estimate and
    returned rc 0 with fewer pairs than the file holds.  An 80 MB file through the one-shot call (several reader
    threads: the pool has at least two) must come back whole; the chunked reader on the same file agrees."""
    rng = np.random.default_rng(7)
    n_lines = 200000
    lens = rng.integers(300, 520, size=n_lines)
    data = rng.integers(65, 91, size=int(lens.sum()) + n_lines, dtype=np.uint8)
    ends = np.cumsum(lens + 1) - 1
    data[ends] = 10
    assert data.size > (72 << 20)
    p = tmp_path / "big64.in"
    with open(p, "wb") as f:
        f.write(b"%d\n" % n_lines)
        f.write(data.tobytes())
    line_num, b, dang = agx.read_sw_text(str(p))
    assert line_num == n_lines and b.n_pairs == n_lines // 2 and dang is None
    assert np.array_equal(b.len.astype(np.int64), lens + 1)
    starts = np.concatenate([[0], ends[:-1] + 1])
    assert np.array_equal(b.off.astype(np.int64), starts)
    for k in (0, 1, n_lines // 2 + 3, n_lines - 1):
        assert b.bases[int(b.off[k]) : int(b.off[k]) + int(b.len[k])].tobytes() == data[starts[k] : ends[k] + 1].tobytes()
    total = 0
    for lnum, cb, d in agx.read_sw_text_chunks(str(p), 30000):
        assert np.array_equal(cb.len.astype(np.int64), lens[2 * total : 2 * (total + cb.n_pairs)] + 1)
        total += cb.n_pairs
    assert total == n_lines // 2


def test_every_tuning_knob_is_listed():
    """The tuning build's knobs (agx_tune("...") in the host sources) are the names api.TUNING_KNOBS selects that build by: a knob
    missing from the list would be set on a process that then loads the shipped library, which reads no environment."""
    import glob, re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "accelerating-genomics_amd", "csrc")
    names = set()
    for f in glob.glob(os.path.join(root, "*.cpp")):
        names |= set(re.findall(r'agx_tune\("([A-Z0-9_]+)"\)', open(f).read()))
    assert names and names <= set(agx.TUNING_KNOBS), sorted(names - set(agx.TUNING_KNOBS))
