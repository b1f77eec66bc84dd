"""-m gpu: parity of the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (fp32 HIP vs fp64 oracle; the reference's release build is itself fp32 + -use_fast_math with
atomics of unspecified order):
  forward tensors / loss   rel 2e-5
  gradients                rel-L2 2e-4 (loss-scale independent), grad-norm rel 1e-4
  parameters after updates rel-L2 1e-4 of the parameter *change*
"""
import numpy as np
import pytest

import cunvsm_amd as ca
from oracle import nvsm_oracle as orc
from tests.helpers import PARAMS, gpu_model, load_params, oracle_model, random_batch, random_params, rel_err

pytestmark = pytest.mark.gpu

FWD_TOL, GRAD_TOL, UPD_TOL = 2e-5, 2e-4, 2e-4


# ---------------------------------------------------------------------------------------------
# unit kernels
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (130, 70, 45), (64, 256, 300), (300, 256, 1000), (5, 3, 2), (257, 300, 256),
                                   # batch-sized products against a projection matrix: the LDS-stationary kernel
                                   # (gemm_tstat.hip; layouts 0 and 1), ragged row counts, both projection shapes
                                   (1031, 256, 300), (4099, 300, 256), (2048, 256, 128), (1500, 128, 256)])
@pytest.mark.parametrize("layout", [0, 1, 2, 3])
def test_gemm_layouts(M, N, K, layout):
    """fp32 MFMA GEMM incl. an asymmetric operand (transposed-output bugs show up, cdna guide G9)."""
    rs = np.random.RandomState(M * 7 + N * 3 + K + layout)
    A = rs.uniform(-1, 1, (M, K)).astype(np.float32)
    Bm = (rs.uniform(-1, 1, (K, N)) + np.arange(N)[None, :] * 0.01).astype(np.float32)
    al, bl = (layout >> 1) & 1, layout & 1
    Ah = np.ascontiguousarray(A.T if al else A)
    Bh = np.ascontiguousarray(Bm.T if bl else Bm)
    Cout = np.empty((M, N), np.float32)
    ca._lib.check(ca.lib().nvsm_debug_gemm(layout, M, N, K, Ah.ctypes.data, Bh.ctypes.data, Cout.ctypes.data))
    ref = A.astype(np.float64) @ Bm.astype(np.float64)
    assert rel_err(Cout, ref) < 2e-6


@pytest.mark.parametrize("M,N,K", [(1031, 256, 300), (4099, 300, 256), (2048, 256, 128), (1500, 128, 256), (16390, 256, 300)])
@pytest.mark.parametrize("layout", [0, 1])
def test_gemm_large_batch_kernels(M, N, K, layout, monkeypatch):
    """The same batch-sized products with the row-panel kernel of the per-rank batch sizes (gemm_rows.hip, which takes
    512 <= M <= 16 384 by default) switched off: the LDS-stationary / tiled kernels the 51 200-window batch runs on."""
    monkeypatch.setenv("NVSM_GEMM_ROWS_MAX", "0")
    test_gemm_layouts(M, N, K, layout)


@pytest.mark.parametrize("M", [1024, 3333, 16390, 51200])      # (51 200: the metric's batch — 200 rows per workgroup, 256 workgroups)
@pytest.mark.parametrize("layout,N,K", [(0, 256, 300), (1, 300, 256), (1, 272, 64)])
def test_gemm_split_bf16_is_fp32_accurate(M, layout, N, K, monkeypatch):
    """gemm_split.hip: fp32 operands cut exactly into three bf16 planes, nine or six bf16 MFMAs per product, fp32 accumulation.
    Against an fp64 product, on operands spread over many binades (as gradients are), its error must not exceed the exact-fp32
    MFMA kernels' by more than rounding noise — and the 9- and 6-product forms must agree to fp32 roundoff."""
    monkeypatch.setenv("NVSM_GEMM_ROWS_MAX", "0")
    rs = np.random.RandomState(M + N + K)
    A = (rs.standard_normal((M, K)) * np.exp2(rs.randint(-12, 4, (M, K)))).astype(np.float32)
    Bm = (rs.standard_normal((K, N)) * 0.1 * np.exp2(rs.randint(-6, 3, (K, N))) + np.arange(N)[None, :] * 1e-3).astype(np.float32)
    Bh = np.ascontiguousarray(Bm.T if layout else Bm)
    ref = A.astype(np.float64) @ Bm.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(Bm).astype(np.float64)
    out, err = {}, {}
    for mode in ("0", "9", "6"):
        monkeypatch.setenv("NVSM_GEMM_SPLIT", mode)
        out[mode] = np.empty((M, N), np.float32)
        ca._lib.check(ca.lib().nvsm_debug_gemm(layout, M, N, K, A.ctypes.data, Bh.ctypes.data, out[mode].ctypes.data))
        e = (out[mode].astype(np.float64) - ref) / scale
        err[mode] = (np.abs(e).max(), np.sqrt((e ** 2).mean()))
    assert not np.array_equal(out["0"], out["9"])                      # (the split kernel did run)
    for mode in ("9", "6"):
        assert err[mode][0] < 1.5 * err["0"][0] + 1e-8, err            # largest error, relative to Σ|a b|
        assert err[mode][1] < 1.1 * err["0"][1] + 1e-9, err            # root mean square
    assert rel_err(out["6"], out["9"].astype(np.float64)) < 3e-7


@pytest.mark.parametrize("M", [512, 1031, 4096, 6400, 8192])
@pytest.mark.parametrize("layout,N,K", [(0, 256, 300), (1, 300, 256), (0, 256, 128), (1, 128, 256), (1, 272, 64), (0, 64, 20)])
def test_gemm_rsplit_bf16_is_fp32_accurate(M, layout, N, K, monkeypatch):
    """gemm_rsplit.hip — the same arithmetic for the per-rank batch sizes (32-row panels, the panel's planes in LDS at once): held
    against the exact-fp32 row-panel kernel (gemm_rows.hip, NVSM_GEMM_SPLIT=0) exactly as the large-batch kernel is above; ragged
    row counts (a last workgroup of 7 rows), K not a multiple of 16, fewer column tiles than waves."""
    rs = np.random.RandomState(M + N + K)
    A = (rs.standard_normal((M, K)) * np.exp2(rs.randint(-12, 4, (M, K)))).astype(np.float32)
    Bm = (rs.standard_normal((K, N)) * 0.1 * np.exp2(rs.randint(-6, 3, (K, N))) + np.arange(N)[None, :] * 1e-3).astype(np.float32)
    Bh = np.ascontiguousarray(Bm.T if layout else Bm)
    ref = A.astype(np.float64) @ Bm.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(Bm).astype(np.float64)
    out, err = {}, {}
    for mode in ("0", "9", "6"):
        monkeypatch.setenv("NVSM_GEMM_SPLIT", mode)
        out[mode] = np.empty((M, N), np.float32)
        ca._lib.check(ca.lib().nvsm_debug_gemm(layout, M, N, K, A.ctypes.data, Bh.ctypes.data, out[mode].ctypes.data))
        e = (out[mode].astype(np.float64) - ref) / scale
        err[mode] = (np.abs(e).max(), np.sqrt((e ** 2).mean()))
    assert not np.array_equal(out["0"], out["9"])                      # (the split kernel did run)
    for mode in ("9", "6"):
        assert err[mode][0] < 1.5 * err["0"][0] + 1e-8, err            # largest error, relative to Σ|a b|
        assert err[mode][1] < 1.1 * err["0"][1] + 1e-9, err            # root mean square
    assert rel_err(out["6"], out["9"].astype(np.float64)) < 3e-7


@pytest.mark.parametrize("split", [2, 7, 128])
@pytest.mark.parametrize("exact", [0, 1])
def test_gemm_split_k(split, exact):
    """The projection gradient's shape: dT = Aᵀ·B split-K over the batch — the split-bf16 kernel (gemm_dt.hip) and the tiled
    exact-fp32 one (variant bit 30)."""
    rs = np.random.RandomState(split)
    M, N, K = 300, 256, 4096
    A = rs.uniform(-1, 1, (K, M)).astype(np.float32)       # stored [K][M] like phrase
    Bm = rs.uniform(-1, 1, (K, N)).astype(np.float32)
    Cout = np.empty((M, N), np.float32)
    ca._lib.check(ca.lib().nvsm_debug_gemm((exact << 30) | (split << 2) | 2, M, N, K, A.ctypes.data, Bm.ctypes.data, Cout.ctypes.data))
    assert rel_err(Cout, A.T.astype(np.float64) @ Bm.astype(np.float64)) < 2e-6


@pytest.mark.parametrize("M,N,K,split", [(300, 256, 51200, 128), (300, 256, 6400 + 17, 50), (64, 200, 1000, 3), (320, 132, 77, 1), (16, 256, 64, 4),
                                         (300, 256, 16, 1), (128, 128, 4096, 256), (300, 256, 12800 + 5, 247), (292, 228, 333, 2), (4, 4, 1, 1)])
@pytest.mark.parametrize("products", ["6", "9"])
def test_gemm_dt_split_bf16(M, N, K, split, products, monkeypatch):
    """gemm_dt.hip on ragged slabs (a last tile of 1 / 5 / 13 rows, a batch shorter than a slab, a batch of one tile, padding
    columns, shapes with and without idle waves: the FULL and the general instantiation, one and two column halves), an
    asymmetric operand, values over many binades; against an fp64 product, relative to Σ|a b|."""
    monkeypatch.setenv("NVSM_GEMM_SPLIT", products)
    rs = np.random.RandomState(M + N + K)
    A = (rs.standard_normal((K, M)) * np.exp2(rs.randint(-10, 3, (K, M)))).astype(np.float32)
    Bm = (rs.standard_normal((K, N)) * np.exp2(rs.randint(-10, 3, (K, N))) + np.arange(N)[None, :] * 1e-3).astype(np.float32)
    Cout = np.empty((M, N), np.float32)
    ca._lib.check(ca.lib().nvsm_debug_gemm((split << 2) | 2, M, N, K, A.ctypes.data, Bm.ctypes.data, Cout.ctypes.data))
    ref = A.T.astype(np.float64) @ Bm.astype(np.float64)
    scale = np.abs(A.T).astype(np.float64) @ np.abs(Bm).astype(np.float64)
    assert np.abs((Cout - ref) / scale).max() < 1.5e-6


@pytest.mark.parametrize("M,N,K,split", [(300, 256, 6400, 16), (300, 256, 6400 + 17, 50), (128, 256, 4096, 10), (64, 200, 1000, 3), (320, 132, 77, 1),
                                         (16, 256, 64, 4), (300, 256, 16, 1), (292, 228, 333, 2), (7, 4, 1, 1), (33, 68, 129, 5)])
@pytest.mark.parametrize("products", ["6", "9"])
def test_gemm_dtw_split_bf16(M, N, K, split, products, monkeypatch):
    """gemm_dtw.hip (round 6: the projection gradient of per-rank batches in workgroups of one wave; replaces cpp/params.cu:526-531)
    on ragged slabs (a last k step of 1 / 5 / 13 rows, a slab whose length is not a multiple of 8, a batch of one row), padding
    columns on both operands, values over many binades; against an fp64 product, relative to Σ|a b| — gemm_dt.hip's test and bound."""
    monkeypatch.setenv("NVSM_GEMM_SPLIT", products)
    rs = np.random.RandomState(M + N + K)
    A = (rs.standard_normal((K, M)) * np.exp2(rs.randint(-10, 3, (K, M)))).astype(np.float32)
    Bm = (rs.standard_normal((K, N)) * np.exp2(rs.randint(-10, 3, (K, N))) + np.arange(N)[None, :] * 1e-3).astype(np.float32)
    Cout = np.full((M, N), np.nan, np.float32)
    ca._lib.check(ca.lib().nvsm_debug_gemm((1 << 29) | (split << 2) | 2, M, N, K, A.ctypes.data, Bm.ctypes.data, Cout.ctypes.data))
    ref = A.T.astype(np.float64) @ Bm.astype(np.float64)
    scale = np.abs(A.T).astype(np.float64) @ np.abs(Bm).astype(np.float64)
    assert np.abs((Cout - ref) / scale).max() < 1.5e-6


def test_gemm_identity_asymmetric():
    M = N = K = 128
    A = np.eye(M, dtype=np.float32)
    Bm = np.arange(K * N, dtype=np.float32).reshape(K, N) / 1000.0
    Cout = np.empty((M, N), np.float32)
    ca._lib.check(ca.lib().nvsm_debug_gemm(0, M, N, K, A.ctypes.data, Bm.ctypes.data, Cout.ctypes.data))
    np.testing.assert_array_equal(Cout, Bm)


@pytest.mark.parametrize("dim,window,weighted", [(3, 3, False), (3, 3, True), (300, 10, True), (256, 1, False), (128, 10, True), (7, 2, True)])
def test_gather_mean(dim, window, weighted):
    rs = np.random.RandomState(dim + window)
    rows, n = 50, 37
    table = rs.uniform(-1, 1, rows * dim).astype(np.float32)
    idx = rs.randint(0, rows, n * window).astype(np.int64)
    w = rs.uniform(0, 2, n * window).astype(np.float32) if weighted else None
    out = np.empty(n * dim, np.float32)
    ca._lib.check(ca.lib().nvsm_debug_gather_mean(rows, dim, table.ctypes.data, idx.ctypes.data,
                                                  None if w is None else w.ctypes.data, window, n, out.ctypes.data))
    ref = orc.average_repr(table, dim, idx, w, window)
    np.testing.assert_allclose(out, ref, rtol=2e-6, atol=1e-7)


def test_gather_mean_reference_kat():
    """cpp/model_tests.cu:52-123 through the HIP kernel."""
    table = np.arange(12, dtype=np.float32)
    idx = np.array([1, 3, 2, 0, 3, 1], np.int64)
    w = np.array([0.5, 0.3, 0.1, 1.0, 2.0, 0.2], np.float32)
    out = np.empty(6, np.float32)
    ca._lib.check(ca.lib().nvsm_debug_gather_mean(4, 3, table.ctypes.data, idx.ctypes.data, None, 3, 2, out.ctypes.data))
    np.testing.assert_allclose(out, [6, 7, 8, 4, 5, 6], rtol=1e-6)
    ca._lib.check(ca.lib().nvsm_debug_gather_mean(4, 3, table.ctypes.data, idx.ctypes.data, w.ctypes.data, 3, 2, out.ctypes.data))
    np.testing.assert_allclose(out, [(0.5 * 3 + 0.3 * 9 + 0.1 * 6) / 3., (0.5 * 4 + 0.3 * 10 + 0.1 * 7) / 3.,
                                     (0.5 * 5 + 0.3 * 11 + 0.1 * 8) / 3., (0 + 2.0 * 9 + 0.2 * 3) / 3.,
                                     (1.0 + 2.0 * 10 + 0.2 * 4) / 3., (2.0 + 2.0 * 11 + 0.2 * 5) / 3.], rtol=1e-6)


GATHER_FUSED_SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
import cunvsm_amd as ca
from tests.helpers import PARAMS, gpu_model, load_params, random_batch, random_params
out = []
for dw, de, window, B, weighted, lazy in ((300, 256, 10, 6400, True, False), (128, 128, 10, 4096, False, False), (60, 64, 6, 1000, True, False),
                                          (300, 256, 3, 8192, True, False), (64, 96, 5, 515, True, False), (64, 96, 5, 515, True, True)):
    spec = dict(num_words=5000, num_entities=700, word_dim=dw, entity_dim=de, window=window, num_random=3, nonlinearity="hard_tanh",
                batch_norm=True, update_method="sparse_adam" if lazy else "sgd", **{"lambda": 0.01})
    rs = np.random.RandomState(B + dw)
    g = gpu_model(spec, B)
    params = random_params(spec, rs)
    load_params(g, params, True)
    desc = g.describe(B)
    if lazy:      # two updates first: the words table then carries pending decay factors the gather has to apply (LazyView)
        for _ in range(2):
            words, ww, labels, iw, ids = random_batch(spec, rs, B, zipf=True, weighted=True)
            g.step(ca.Batch(words, labels, ww, iw), 0.05, entity_ids=ids)
    words, ww, labels, iw, ids = random_batch(spec, rs, B, zipf=True, weighted=weighted)
    g.profile_enable(True)
    g.compute_cost(ca.Batch(words, labels, ww if weighted else None, iw), ids)
    g.synchronize()
    notes = sorted(g.profile())
    phrase = g.get_tensor("phrase")
    pre = g.get_tensor("pre").reshape(B, de)
    table = g.get_param(PARAMS[0])           # (flushes pending decay: the rows as the dense passes would have left them)
    T = g.get_param(PARAMS[2]).reshape(dw, de).astype(np.float64)
    ref_phrase = np.empty(B * dw, np.float32)
    ca._lib.check(ca.lib().nvsm_debug_gather_mean(spec["num_words"], dw, table.ctypes.data, words.ctypes.data,
                                                  ww.ctypes.data if weighted else None, window, B, ref_phrase.ctypes.data))
    ref = ref_phrase.reshape(B, dw).astype(np.float64) @ T
    out.append(dict(shape=[dw, de, window, B, weighted, lazy], desc=desc, notes=notes,
                    phrase_bits_equal=bool(np.array_equal(phrase.view(np.uint32), ref_phrase.view(np.uint32))),
                    pre_err=float(np.abs(pre - ref).max() / np.abs(ref).max())))
print("RESULT " + json.dumps(out))
"""


def test_gather_inside_the_forward_product_writes_the_gather_kernels_bits():
    """Round 6 experiment (NVSM_GATHER_FUSE=1, experiments build; off by default because it measured slower, profiles/NOTES_r06.md §1):
    at per-rank batch sizes the forward product forms the phrase rows itself while it stages them (gemm_rsplit.hip GATH; replaces
    cpp/params.cu:75-95 + :417 as one launch). The `phrase` side output must be the stand-alone gather kernel's bits
    (nvsm_debug_gather_mean on the same table, ids and weights) — ragged last panel, no weights, lazily decayed table included."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dbg = os.path.join(root, "cunvsm_amd", "libcunvsm_amd_dbg.so")
    if not os.path.exists(dbg):
        pytest.skip("experiments build absent (make -C cunvsm_amd/csrc dbg)")
    env = dict(os.environ, CUNVSM_AMD_LIB=dbg, NVSM_GATHER_FUSE="1", NVSM_LAZY_MIN_MB="0")
    r = subprocess.run([sys.executable, "-c", GATHER_FUSED_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, cwd=root, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-2000:])
    for e in json.loads(lines[-1][len("RESULT "):]):
        assert "with the word gather inside" in e["desc"], e
        assert "gather_in_product" in e["notes"] and "gather_mean_words" not in e["notes"], e
        assert e["phrase_bits_equal"], e
        assert e["pre_err"] <= 2e-5, e


# ---------------------------------------------------------------------------------------------
# initialisation + sampling replay the reference's RNG stream draw for draw
# ---------------------------------------------------------------------------------------------
def test_initialize_and_host_sampler_match_reference_rng():
    spec = dict(num_words=40, num_entities=30, word_dim=6, entity_dim=8, window=3, num_random=4)
    g = gpu_model(spec, 16, sampler=ca.SAMPLER_HOST_MINSTD)
    g.initialize(7)
    rng = orc.Rng(7)
    o = oracle_model(spec, orc.F32)
    o.initialize(rng)
    for name in PARAMS:
        np.testing.assert_array_equal(g.get_param(name), o.get(name).astype(np.float32))
    assert g.rng_state == rng.state
    rs = np.random.RandomState(0)
    words, ww, labels, iw, _ = random_batch(spec, rs, 16)
    g.compute_cost(ca.Batch(words, labels, ww, iw))
    expect = rng.generate_labels(labels, 30, 4)
    np.testing.assert_array_equal(g.get_tensor("entity_ids").astype(np.int64), expect)
    assert g.rng_state == rng.state


@pytest.mark.parametrize("num_entities", [1, 2, 3, 7, 1000, 65536, 100000, 1234567, 16777217])
@pytest.mark.parametrize("seed", [1, 2147483646])
def test_host_sampler_replays_std_uniform_int_distribution(num_entities, seed):
    """The engine's written-out replay of `std::uniform_int_distribution<long>(0, |D|-1)(minstd_rand0)` (model.cpp:
    draw_reference_negatives) against the oracle, which makes the std:: calls themselves: same ids and the same generator
    state afterwards, over table sizes that exercise the rejection loop and the quotient's rounding (three batches each)."""
    spec = dict(num_words=5, num_entities=num_entities, word_dim=1, entity_dim=1, window=1, num_random=20)
    B = 257
    g = gpu_model(spec, B, sampler=ca.SAMPLER_HOST_MINSTD)
    g.initialize(3)
    g.rng_state = seed
    rng = orc.Rng(3)
    rng.state = seed
    rs = np.random.RandomState(num_entities % 1000)
    for _ in range(3):
        words, ww, labels, iw, _ = random_batch(spec, rs, B)
        g.compute_cost(ca.Batch(words, labels, ww, iw))
        expect = rng.generate_labels(labels, num_entities, 20)
        got = g.get_tensor("entity_ids").astype(np.int64)
        if num_entities <= (1 << 24):                    # get_tensor hands ids back as float32: exact up to 2^24
            np.testing.assert_array_equal(got, expect)
        assert g.rng_state == rng.state


@pytest.mark.parametrize("num_entities", [7, 100000])
def test_host_sampler_data_parallel_slices_replay_the_global_batch(num_entities):
    """world_size > 1 with the host sampler (include/cunvsm_amd.h, NEGATIVES): every rank starts from the SAME generator
    state (the trainer hands it over) and holds instances [r·B, (r+1)·B) of the global batch. Rank r's negatives must be
    those the single-GPU run draws for these instances — not the same set on every rank — and every rank's generator must
    end where the single-GPU run's ends (it also shuffles the data source)."""
    spec = dict(num_words=5, num_entities=num_entities, word_dim=1, entity_dim=1, window=1, num_random=6)
    G, B = 3, 130
    one = gpu_model(spec, G * B, sampler=ca.SAMPLER_HOST_MINSTD)
    ranks = [gpu_model(spec, B, sampler=ca.SAMPLER_HOST_MINSTD, world_size=G, rank=r, sync_batch_norm=0) for r in range(G)]
    rs = np.random.RandomState(5)
    state = 12345
    for _ in range(2):
        words, ww, labels, iw, _ = random_batch(spec, rs, G * B)
        one.rng_state = state
        one.compute_cost(ca.Batch(words, labels, ww, iw))
        want = one.get_tensor("entity_ids").astype(np.int64).reshape(G, B, -1)
        for r, m in enumerate(ranks):
            m.rng_state = state
            sl = slice(r * B, (r + 1) * B)
            m.compute_cost(ca.Batch(words[sl], labels[sl], ww[sl], iw[sl]))
            np.testing.assert_array_equal(m.get_tensor("entity_ids").astype(np.int64).reshape(B, -1), want[r])
            assert m.rng_state == one.rng_state
        assert len({want[r][:, 1:].tobytes() for r in range(G)}) == G          # no two ranks share a negative set
        state = one.rng_state


def test_page_locked_source_at_an_odd_offset():
    """A host batch that is a slice of a page-locked buffer starting 8 / 4 bytes into it (a data-parallel rank's share at an
    odd instance offset): only 16-byte aligned sources go through the PCIe pull kernel, anything else through the copy
    engine — same step either way."""
    spec = dict(SPECS["nvsm"], update_method="sgd")
    B = 257
    rs = np.random.RandomState(3)
    params = random_params(spec, rs)
    a, b = gpu_model(spec, B), gpu_model(spec, B)
    for m in (a, b):
        load_params(m, params, True)
    words, ww, labels, iw, ids = random_batch(spec, rs, B)
    w = spec["window"]
    pins = [ca.model.pinned_copy(np.concatenate([x[:1], x])) for x in (words, labels, ww, iw)]       # one element of padding in front
    off = ca.Batch(pins[0].array[1:], pins[1].array[1:], pins[2].array[1:], pins[3].array[1:])
    assert off.features.ctypes.data % 16 == 8 and off.feature_weights.ctypes.data % 16 == 4 and off.features.size == B * w
    ca_cost = a.step(ca.Batch(words, labels, ww, iw), 1e-2, entity_ids=ids, want_cost=True)
    cb_cost = b.step(off, 1e-2, entity_ids=ids, want_cost=True)
    assert ca_cost == cb_cost
    for n in PARAMS:
        np.testing.assert_array_equal(a.get_param(n), b.get_param(n))


# ---------------------------------------------------------------------------------------------
# the reference's own end-to-end KAT (cpp/model_tests.cu:341-466) through the HIP path
# ---------------------------------------------------------------------------------------------
def test_transform_backward_golden(golden):
    gd = golden("transform_backward")
    spec = dict(num_words=gd["num_words"], num_entities=gd["num_entities"], word_dim=gd["word_dim"],
                entity_dim=gd["entity_dim"], window=gd["window_size"], num_random=gd["num_random_entities"],
                bias_negative_samples=True, clip_sigmoid=False, lambda_=gd["regularization_lambda"])
    B, w = gd["batch_size"], gd["window_size"]
    m = gpu_model(spec, B, sampler=ca.SAMPLER_HOST_MINSTD)
    m.initialize(gd["seed"])
    batch = ca.Batch(np.full(B * w, gd["feature_value"]), np.full(B, gd["label"]), np.ones(B * w), np.ones(B))
    m.compute_cost(batch)
    m.get_cost()
    m.compute_gradients()
    # fp32 against fp64 goldens
    np.testing.assert_allclose(m.get_tensor("grad_transform"), gd["grad_transform"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(m.get_tensor("grad_bias"), gd["grad_bias"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(m.get_tensor("grad_phrase"), gd["grad_phrase"], rtol=5e-4, atol=2e-7)


# ---------------------------------------------------------------------------------------------
# forward / backward parity on random models
# ---------------------------------------------------------------------------------------------
SPECS = {
    # LSE recipe shape (tanh, no BN, bias_negative_samples) — scripts/functions.sh:267
    "lse": dict(num_words=500, num_entities=200, word_dim=128, entity_dim=256, window=10, num_random=16,
                nonlinearity="tanh", batch_norm=False, bias_negative_samples=True, lambda_=0.01),
    # NVSM recipe shape (hard_tanh + BN, reweighted negatives) — scripts/functions.sh:266
    "nvsm": dict(num_words=500, num_entities=300, word_dim=300, entity_dim=256, window=10, num_random=16,
                 nonlinearity="hard_tanh", batch_norm=True, lambda_=0.01),
    # the gradient-check fixture of the reference (tests_base_cuda.h:196-200): tiny odd dims
    "tiny": dict(num_words=20, num_entities=15, word_dim=3, entity_dim=4, window=3, num_random=1,
                 nonlinearity="tanh", batch_norm=True, lambda_=0.01),
    "tiny_odd": dict(num_words=20, num_entities=15, word_dim=5, entity_dim=7, window=2, num_random=3,
                     nonlinearity="hard_tanh", batch_norm=False, lambda_=0.0),
    "wide": dict(num_words=64, num_entities=64, word_dim=64, entity_dim=512, window=4, num_random=2,
                 nonlinearity="tanh", batch_norm=True, lambda_=0.0),
    # kernel-dispatch corners: > 64 candidates per example (generic loss kernel), the reference's own k = 10 and a small
    # k (loss_rows_kernel<11> / <6>), window 1, the largest supported document dimension (1024, 4 column slices per lane)
    "many_negatives": dict(num_words=80, num_entities=90, word_dim=32, entity_dim=64, window=3, num_random=69,
                           nonlinearity="tanh", batch_norm=False, lambda_=0.01),
    "k10": dict(num_words=300, num_entities=150, word_dim=300, entity_dim=256, window=10, num_random=10,
                nonlinearity="hard_tanh", batch_norm=True, lambda_=0.01),
    "k4_window1": dict(num_words=50, num_entities=40, word_dim=24, entity_dim=128, window=1, num_random=4,
                       nonlinearity="tanh", batch_norm=True, bias_negative_samples=True, lambda_=0.01),
    # tables much larger than the batch: the row passes go by the touched-row list, the other rows through the
    # streaming dense-decay pass (update.hip: row_pass_split)
    "sparse_touch": dict(num_words=30000, num_entities=40000, word_dim=16, entity_dim=12, window=3, num_random=4,
                         nonlinearity="hard_tanh", batch_norm=True, lambda_=0.01),
    "sparse_touch_odd": dict(num_words=9000, num_entities=7000, word_dim=7, entity_dim=5, window=2, num_random=3,
                             nonlinearity="tanh", batch_norm=False, lambda_=0.01),
    # the optional L2 normalisers (gradient_checking_tests.cu:92-110: phrase / entity / both)
    "l2_phrase": dict(num_words=60, num_entities=40, word_dim=24, entity_dim=20, window=3, num_random=4,
                      nonlinearity="tanh", batch_norm=False, l2_phrase=True, lambda_=0.01),
    "l2_entity": dict(num_words=60, num_entities=40, word_dim=24, entity_dim=20, window=3, num_random=4,
                      nonlinearity="hard_tanh", batch_norm=True, l2_entity=True, lambda_=0.01),
    "l2_both": dict(num_words=300, num_entities=500, word_dim=300, entity_dim=256, window=10, num_random=16,
                    nonlinearity="hard_tanh", batch_norm=True, l2_phrase=True, l2_entity=True, lambda_=0.01),
    "dim1024": dict(num_words=40, num_entities=30, word_dim=16, entity_dim=1024, window=2, num_random=3,
                    nonlinearity="hard_tanh", batch_norm=False, lambda_=0.0),
}
SPECS["nvsm_dw260"] = dict(SPECS["nvsm"], word_dim=260)
SPECS["nvsm_dw364"] = dict(SPECS["nvsm"], word_dim=364, num_random=8)
for _s in SPECS.values():
    _s["lambda"] = _s.pop("lambda_")


def _pair(spec, B, seed, max_batch=None):
    rs = np.random.RandomState(seed)
    params = random_params(spec, rs)
    o = oracle_model(spec)
    g = gpu_model(spec, max_batch or B)
    load_params(o, params, False)
    load_params(g, params, True)
    return o, g, rs


@pytest.mark.parametrize("name,B", [("lse", 256), ("nvsm", 1024), ("tiny", 1024), ("tiny_odd", 100), ("wide", 130), ("nvsm", 1000),
                                    ("many_negatives", 96), ("k10", 512), ("k4_window1", 200), ("dim1024", 64),
                                    ("l2_phrase", 256), ("l2_entity", 256), ("l2_both", 512),
                                    # batches above 8 192 rows: the split-bf16 projection kernels (ragged row blocks, a last
                                    # block of 8 rows, 17 / 23 column blocks in the backward product), same tolerances
                                    ("nvsm", 8200), ("nvsm", 9999), ("nvsm_dw260", 8300), ("nvsm_dw364", 8208)])
def test_forward_backward_parity(name, B):
    spec = SPECS[name]
    o, g, rs = _pair(spec, B, 11)
    words, ww, labels, iw, ids = random_batch(spec, rs, B, zipf=True)
    o.forward(words, ww, ids, iw)
    g.compute_cost(ca.Batch(words, labels, ww, iw), ids)
    co, cg = o.get_cost(), g.get_cost()
    assert abs(cg - co) <= FWD_TOL * abs(co), (cg, co)
    for t in ("phrase", "pre", "proj", "probs"):
        assert rel_err(g.get_tensor(t), o.get(t)) < FWD_TOL, t
    if spec.get("batch_norm"):
        assert rel_err(g.get_tensor("bn_inv_std"), o.get("bn_inv_std")) < FWD_TOL
    o.backward()
    g.compute_gradients()
    for gt, ot in (("grad_transform", "grad_transform"), ("grad_bias", "grad_bias"), ("grad_phrase", "grad_phrase"),
                   ("grad_entity", "grad_entity"), ("grad_proj", "grad_proj")):
        a, b = g.get_tensor(gt), o.get(ot)
        assert rel_err(a, b) < GRAD_TOL, (gt, rel_err(a, b))
        # (norms in fp64: numpy sums a float32 array in float32, and 36 M squares of a large-batch tensor lose four digits that way)
        assert abs(np.linalg.norm(a.astype(np.float64)) - np.linalg.norm(b)) <= 1e-4 * np.linalg.norm(b), gt
    # signed multipliers: oracle keeps |m| and negates the entity rows instead
    R = spec["num_random"] + 1
    sign = np.where(np.arange(B * R) % R == 0, 1.0, -1.0)
    assert rel_err(g.get_tensor("multipliers"), o.get("multipliers") * sign) < GRAD_TOL


def test_split_kernels_follow_a_rewritten_projection():
    """The split-bf16 projection kernels read T as bf16 planes that are cut when T changes (behind the projection update,
    off the critical path): anything else that writes T — set_param, initialize — must leave them stale-marked. A handle
    that has trained, then has its parameters replaced, must compute exactly what a fresh handle with those parameters does."""
    spec = SPECS["nvsm"]
    B = 8300
    rs = np.random.RandomState(3)
    words, ww, labels, iw, ids = random_batch(spec, rs, B, zipf=True)
    batch = ca.Batch(words, labels, ww, iw)
    used = gpu_model(spec, B)
    load_params(used, random_params(spec, rs), True)
    for _ in range(2):
        used.step(batch, 0.05, entity_ids=ids)                 # the planes now hold the trained projection
    params = random_params(spec, rs)
    fresh = gpu_model(spec, B)
    for m in (used, fresh):
        load_params(m, params, True)                           # set_param
        m.compute_cost(batch, ids)
        m.compute_gradients()
    for t in ("pre", "grad_phrase", "grad_transform"):
        np.testing.assert_array_equal(used.get_tensor(t), fresh.get_tensor(t))
    used.step(batch, 0.05, entity_ids=ids)
    used.initialize(9)                                         # initialize
    fresh.initialize(9)
    for m in (used, fresh):
        m.compute_cost(batch, ids)
    np.testing.assert_array_equal(used.get_tensor("pre"), fresh.get_tensor("pre"))


@pytest.mark.parametrize("method", ["sgd", "adagrad", "sparse_adam", "dense_adam", "full_adam"])
@pytest.mark.parametrize("name", ["nvsm", "tiny", "lse", "tiny_odd", "k4_window1", "many_negatives", "sparse_touch", "sparse_touch_odd",
                                  "l2_phrase", "l2_entity", "l2_both"])
@pytest.mark.parametrize("lam", [0.0, 0.01])
def test_update_parity(method, name, lam):
    """Three optimiser steps on identical batches; parameters and optimiser state must track the oracle."""
    spec = dict(SPECS[name], update_method=method)
    spec["lambda"] = lam
    B = 512
    o, g, rs = _pair(spec, B, 5)
    lr = {"sgd": 0.1, "adagrad": 0.01}.get(method, 0.001)      # tests_base_cuda.h:117-130
    start = {p: o.get(p).copy() for p in PARAMS}
    for step in range(3):
        words, ww, labels, iw, ids = random_batch(spec, rs, B, zipf=True)
        o.forward(words, ww, ids, iw)
        o.backward()
        o.update(lr)
        g.compute_cost(ca.Batch(words, labels, ww, iw), ids)
        g.compute_gradients()
        g.update(lr)
        assert abs(g.get_cost() - o.get_cost()) <= 5e-5 * abs(o.get_cost()), step
    for p in PARAMS:
        delta = np.linalg.norm(o.get(p) - start[p])
        err = np.linalg.norm(g.get_param(p).astype(np.float64) - o.get(p))
        # + the fp32 storage floor: three successive roundings of every parameter (half an ulp = 6e-8 relative each)
        assert err <= UPD_TOL * max(delta, 1e-12) + 2e-7 * np.linalg.norm(o.get(p)), (p, err, delta)
    if method != "sgd":
        pairs = {"adagrad": [("word_representations/a", "words.s0"), ("entity_representations/a", "entities.s0"),
                             ("word_entity_mapping/s0_transform", "transform.s0.transform")],
                 }.get(method, [("word_representations/m", "words.s0"), ("entity_representations/m", "entities.s0"),
                                ("word_representations/v", "words.s1"), ("entity_representations/v", "entities.s1"),
                                ("word_entity_mapping/s0_bias", "transform.s0.bias"),
                                ("word_entity_mapping/s1_transform", "transform.s1.transform")])
        for gn, on in pairs:
            assert rel_err(g.get_param(gn), o.get(on)) < 5e-4, gn


@pytest.mark.parametrize("method", ["sgd", "adagrad", "sparse_adam", "dense_adam", "full_adam"])
def test_hot_rows_are_chunked(method):
    """Zipf head: a few words receive thousands of updates (rows longer than the 128-entry chunk go through
    the two-level reduction), the tail a handful; entities also exceed one chunk. One step, so the comparison
    is not amplified by the (huge) hot-row updates of the reference's sparse-Adam rule."""
    spec = dict(num_words=40, num_entities=30, word_dim=300, entity_dim=256, window=10, num_random=4,
                nonlinearity="hard_tanh", batch_norm=True, update_method=method)
    spec["lambda"] = 0.01
    B = 1024
    o, g, rs = _pair(spec, B, 3)
    start = {p: o.get(p).copy() for p in PARAMS}
    words, ww, labels, iw, ids = random_batch(spec, rs, B, zipf=True)
    assert np.bincount(words, minlength=40).max() > 1500 and np.bincount(ids, minlength=30).max() > 128
    lr = {"sgd": 0.1, "adagrad": 0.01}.get(method, 0.001)
    o.forward(words, ww, ids, iw); o.backward(); o.update(lr)
    g.compute_cost(ca.Batch(words, labels, ww, iw), ids); g.compute_gradients(); g.update(lr)
    for p in PARAMS:
        delta = np.linalg.norm(o.get(p) - start[p])
        err = np.linalg.norm(g.get_param(p).astype(np.float64) - o.get(p))
        assert err <= UPD_TOL * delta + 1e-7 * np.linalg.norm(o.get(p)), (p, err, delta)


@pytest.mark.parametrize("method", ["sgd", "adagrad", "sparse_adam", "dense_adam", "full_adam"])
@pytest.mark.parametrize("dims", [(300, 256), (7, 5)])
def test_one_launch_table_pass_equals_three_launches(method, dims):
    """The update's table passes in one launch (chunk tree of the long rows handed over by last-arriver counters, update.hip
    table_pass_kernel) against the chunk / level-2 / row launches: same additions in the same order, so every parameter
    and optimiser state is bit-identical. 40 words under Zipf at batch 4096: the head rows hold > 4096 entries (both
    levels of the tree), the tail a handful; three steps, so that counters left non-zero by a pass would show."""
    spec = dict(num_words=40, num_entities=30, word_dim=dims[0], entity_dim=dims[1], window=10, num_random=4,
                nonlinearity="hard_tanh", batch_norm=True, update_method=method)
    spec["lambda"] = 0.01
    B = 4096
    rs = np.random.RandomState(11)
    params = random_params(spec, rs)
    batches = [random_batch(spec, rs, B, zipf=True) for _ in range(3)]
    assert np.bincount(batches[0][0], minlength=40).max() > 4096
    results = []
    try:
        for one_launch in (1, 0):
            ca._lib.check(ca.lib().nvsm_debug_set_table_pass_form(one_launch))
            g = gpu_model(spec, B)
            load_params(g, params, True)
            for words, ww, labels, iw, ids in batches:
                g.compute_cost(ca.Batch(words, labels, ww, iw), ids); g.compute_gradients(); g.update(0.001)
            state = {"adagrad": ["word_representations/a", "entity_representations/a"],
                     "sparse_adam": ["word_representations/m", "word_representations/v", "entity_representations/m",
                                     "entity_representations/v"],
                     "dense_adam": ["word_representations/m", "word_representations/v", "entity_representations/m",
                                    "entity_representations/v"],
                     "full_adam": ["word_representations/m", "word_representations/v", "entity_representations/m",
                                   "entity_representations/v"]}.get(method, [])
            results.append({p: g.get_param(p) for p in list(PARAMS) + state})
    finally:
        ca._lib.check(ca.lib().nvsm_debug_set_table_pass_form(1))
    for p in results[0]:
        np.testing.assert_array_equal(results[0][p], results[1][p], err_msg=p)


def test_edge_cases_ragged_and_minimal():
    """B not a multiple of anything, B = 1, k = 0 (no negatives), window = 1, NULL weights."""
    spec = dict(num_words=30, num_entities=20, word_dim=8, entity_dim=12, window=1, num_random=0,
                nonlinearity="tanh", batch_norm=False, bias_negative_samples=True, update_method="sgd")
    spec["lambda"] = 0.0
    for B in (1, 3, 65):
        o, g, rs = _pair(spec, B, B, max_batch=65)
        words, _, labels, _, ids = random_batch(spec, rs, B, weighted=False)
        o.forward(words, np.ones(B), ids, np.ones(B))
        g.compute_cost(ca.Batch(words, labels), ids)          # NULL feature_weights / weights = 1.0
        assert abs(g.get_cost() - o.get_cost()) <= FWD_TOL * abs(o.get_cost())
        o.backward(); g.compute_gradients()
        assert rel_err(g.get_tensor("grad_transform"), o.get("grad_transform")) < GRAD_TOL
        o.update(0.1); g.update(0.1)
        for p in PARAMS:
            assert rel_err(g.get_param(p), o.get(p)) < 1e-5, p


def test_error_behaviour():
    spec = dict(num_words=30, num_entities=20, word_dim=8, entity_dim=12, window=2, num_random=1)
    g = gpu_model(spec, 8)
    with pytest.raises(ca.NvsmError) as e:
        g.compute_gradients()
    assert e.value.status == 4
    with pytest.raises(ca.NvsmError) as e:
        g.compute_cost(ca.Batch(np.zeros(2 * 9, np.int64), np.zeros(9, np.int64)))     # over capacity
    assert e.value.status == 1
    with pytest.raises(ca.NvsmError):
        g.initialize(0)                                                                 # cpp/main.cu:708


def test_deferred_cost_equals_immediate_cost():
    """nvsm_step_deferred + nvsm_deferred_cost (the trainer's loop: the loss of batch k is read after batch k+1 has been
    queued) return what nvsm_step returns, and leave the same parameters."""
    spec = dict(SPECS["nvsm"], update_method="sparse_adam")
    B = 512
    o, a, rs = _pair(spec, B, 21)
    b = gpu_model(spec, B)
    for n in PARAMS:
        b.set_param(n, a.get_param(n))
    batches = [random_batch(spec, rs, B, zipf=True) for _ in range(5)]
    want = [a.step(ca.Batch(w, l, ww, iw), 1e-3, entity_ids=ids, want_cost=True) for (w, ww, l, iw, ids) in batches]
    tickets = []
    for (w, ww, l, iw, ids) in batches:
        b.wait_inputs()
        tickets.append(b.step_deferred(ca.Batch(w, l, ww, iw), 1e-3, entity_ids=ids))
    got = [b.deferred_cost(t) for t in tickets]
    assert got == want
    for n in PARAMS:
        np.testing.assert_array_equal(a.get_param(n), b.get_param(n))
    with pytest.raises(ca.NvsmError):
        b.deferred_cost(tickets[-1] + 1)                  # never issued
    for (w, ww, l, iw, ids) in batches * 2:               # ten more steps: the first tickets fall out of the window
        b.step_deferred(ca.Batch(w, l, ww, iw), 1e-3, entity_ids=ids)
    with pytest.raises(ca.NvsmError):
        b.deferred_cost(tickets[0])


def test_mixed_host_and_device_batches_ragged_sizes_back_to_back():
    """Thirty fused steps queued without a host wait — host batches (page-locked and pageable) and device-resident
    batches interleaved, batch sizes changing from step to step — leave exactly the parameters that separate, fully
    ordered calls leave (double-buffered host staging on the copy stream, deferred side-stream joins, ragged batches)."""
    import torch
    spec = dict(SPECS["nvsm"], update_method="sparse_adam")
    Bmax = 1536
    rs = np.random.RandomState(99)
    params = random_params(spec, rs)
    a, b = gpu_model(spec, Bmax, sampler=ca.SAMPLER_DEVICE), gpu_model(spec, Bmax, sampler=ca.SAMPLER_DEVICE)
    for m in (a, b):
        m.initialize(11)
        load_params(m, params, True)
    dev = torch.device("cuda", 0)
    plan, keep = [], []
    for s in range(30):
        B = int(rs.choice([1, 7, 256, 1000, 1024, 1536]))
        words, ww, labels, iw, _ = random_batch(spec, rs, B, zipf=True)
        kind = s % 3
        if kind == 0:
            fused = ca.Batch(words, labels, ww, iw)                                   # pageable host memory
        elif kind == 1:
            pins = [ca.model.pinned_copy(x) for x in (words, labels, ww, iw)]
            keep.append(pins)
            fused = ca.Batch(pins[0].array, pins[1].array, pins[2].array, pins[3].array)
        else:
            fused = ca.Batch(torch.from_numpy(words).to(dev), torch.from_numpy(labels).to(dev),
                             torch.from_numpy(ww).to(dev), torch.from_numpy(iw).to(dev))
        plan.append((fused, ca.Batch(words, labels, ww, iw)))
    tickets = [b.step_deferred(fused, 1e-3) for fused, _ in plan]
    costs_b = [b.deferred_cost(t) for t in tickets[-8:]]
    costs_a = []
    for _, host in plan:
        a.compute_cost(host)
        a.compute_gradients()
        costs_a.append(a.get_cost())
        a.update(1e-3)
    assert costs_b == costs_a[-8:]
    for n in PARAMS:
        np.testing.assert_array_equal(a.get_param(n), b.get_param(n))

