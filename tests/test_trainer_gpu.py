"""GPU tests of the cuNVSMTrainModel replacement (cunvsm_amd/host/train_main.cpp) and of the ABI entry points it adds:
BASELINE.json configs[0] — LSE on the Cranfield collection, batch 4096, tanh — end to end through the C ABI."""
import os
import re
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT
from tests.test_host_layer import H5DUMP, h5_header, parse_metadata

pytestmark = pytest.mark.gpu

TRAINER = os.path.join(ROOT, "cunvsm_amd", "bin", "cuNVSMTrainModel")
CRANFIELD = os.path.join(ROOT, "tests", "golden", "cranfield", "cranfield.trectext")
# scripts/functions.sh:380-399 + the LSE line of :266-267
LSE_ARGS = ["--word_repr_size", "300", "--entity_repr_size", "256", "--window_size", "10", "--num_random_entities", "10",
            "--regularization_lambda", "1e-2", "--learning_rate", "1e-3", "--weighting", "uniform", "--seed", "1",
            "--update_method", "full_adam", "--batch_size", "4096", "--nonlinearity", "tanh", "--bias_negative_samples",
            "--max_vocabulary_size", "65536", "--min_document_frequency", "0"]


def run_trainer(args, timeout=600):
    if not os.path.exists(TRAINER):
        pytest.fail("%s is missing: __graft_entry__.build() builds it" % TRAINER)
    r = subprocess.run([TRAINER] + args, capture_output=True, text=True, timeout=timeout)
    return r


def epoch_costs(stderr):
    last = [l for l in stderr.splitlines() if re.search(r"Epoch #\d+.*cost=\[", l)][-1]
    return [float(x) for x in re.search(r"cost=\[(.*)\]", last).group(1).split(",") if x.strip()]


def test_lse_on_cranfield(tmp_path):
    out = str(tmp_path / "lse")
    r = run_trainer(LSE_ARGS + ["--num_epochs", "3", "--compute_initial_cost", "--dump_initial_model", "--output", out, CRANFIELD])
    assert r.returncode == 0, r.stderr[-3000:]
    costs = epoch_costs(r.stderr)
    assert len(costs) == 4                                   # initial + 3 epochs
    # 11 candidates under bias_negative_samples: the untrained cost is (k+1)·ln 2 per instance
    assert abs(costs[0] - 11 * np.log(2.0)) < 0.5          # Glorot-initialised projections are small, not zero
    assert costs[3] < costs[2] < costs[1] < costs[0]
    assert "Skipping Batch #" in r.stderr                    # the ragged last batch is skipped, as the reference does
    m = re.search(r"vocabulary size=(\d+), corpus size=(\d+)", r.stderr)
    nV, nD = int(m.group(1)), int(m.group(2))
    assert nD == 1398
    meta = parse_metadata(out + "_meta")
    assert len(meta.term) == nV and len(meta.object) == nD
    assert sorted(t.model_term_id for t in meta.term) == list(range(nV))
    assert [o.index_object_id for o in meta.object][:3] == [1, 2, 3]
    for epoch in range(4):
        path = "%s_%d.hdf5" % (out, epoch)
        assert os.path.exists(path), path
        assert h5_header(path) == {"entity_representations-representations": ("H5T_IEEE_F32LE", (nD, 256)),
                                   "word_entity_mapping-bias": ("H5T_IEEE_F32LE", (1, 256)),
                                   "word_entity_mapping-transform": ("H5T_IEEE_F32LE", (300, 256)),
                                   "word_representations-representations": ("H5T_IEEE_F32LE", (nV, 300))}
    # the initial dump holds the Glorot initialisation: bias all zero, |W| bounded by sqrt(6 / (dim + rows))
    bias = subprocess.run([H5DUMP, "-d", "word_entity_mapping-bias", "-y", "-w", "0", out + "_0.hdf5"], capture_output=True, text=True).stdout
    vals = [float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", bias.split("DATA {")[1])]
    assert len(vals) == 256 and all(v == 0.0 for v in vals)


def test_trainer_is_deterministic(tmp_path):
    """Same seed ⇒ the same cost trajectory, bit for bit: host sampler draw-for-draw, sorted (atomic-free) scatter."""
    args = LSE_ARGS + ["--num_epochs", "1", "--document_cutoff", "300", CRANFIELD]
    a, b = run_trainer(args), run_trainer(args)
    assert a.returncode == 0 and b.returncode == 0, a.stderr[-2000:]
    assert epoch_costs(a.stderr) == epoch_costs(b.stderr)


def test_nvsm_recipe_runs(tmp_path):
    """NVSM flags of scripts/functions.sh:266 (hard_tanh + batch normalisation, no negative-sample bias), ragged batches allowed."""
    args = ["--word_repr_size", "64", "--entity_repr_size", "32", "--window_size", "10", "--num_random_entities", "4", "--seed", "1",
            "--update_method", "sparse_adam", "--batch_size", "2048", "--nonlinearity", "hard_tanh", "--batch_normalization",
            "--num_epochs", "2", "--allow_ragged_batches", "--sampler", "device", "--v", "1", "--dump_every", "30",
            "--output", str(tmp_path / "nvsm"), CRANFIELD]
    r = run_trainer(args)
    assert r.returncode == 0, r.stderr[-3000:]
    costs = epoch_costs(r.stderr)
    assert len(costs) == 2 and costs[1] < costs[0]
    # --dump_every (cpp/main.cu:454-459): "<output>_<epoch>_<batch>.hdf5" every 30 batches + "<output>_<epoch>.hdf5" per epoch
    for name in ("nvsm_1_30.hdf5", "nvsm_1_60.hdf5", "nvsm_1.hdf5", "nvsm_2_30.hdf5", "nvsm_2.hdf5", "nvsm_meta"):
        assert os.path.exists(str(tmp_path / name)), name
    assert not os.path.exists(str(tmp_path / "nvsm_0.hdf5"))          # no --dump_initial_model
    assert "Skipping Batch" not in r.stderr
    assert re.search(r"Batch #0 .*cost=", r.stderr)


def test_trainer_refuses_what_it_cannot_do(tmp_path):
    r = run_trainer(["--update_method", "sgd", "--nonlinearity", "tanh", "--seed", "1", "--entity_similarity_weight", "0.5", CRANFIELD])
    assert r.returncode == 1 and "only the text-entity objective" in r.stderr
    r = run_trainer(["--update_method", "sgd", "--nonlinearity", "tanh", "--seed", "1", str(tmp_path)])
    assert r.returncode == 1 and "Unable to open Indri parameters" in r.stderr
    r = run_trainer(["--update_method", "sgd", "--nonlinearity", "tanh", CRANFIELD])
    assert r.returncode == 1 and "Please specify a --seed value." in r.stderr
    r = run_trainer(["--update_method", "nope", "--nonlinearity", "tanh", "--seed", "1", CRANFIELD])
    assert r.returncode == 1 and "Please specify a valid --update_method." in r.stderr


def test_trains_from_the_indri_repository(tmp_path):
    """The reference's own input format: the Indri 5.8 repository it ships for its tests (500 Brown-corpus documents)."""
    brown = os.path.join(ROOT, "tests", "golden", "Brown_index")
    out = str(tmp_path / "brown")
    r = run_trainer(["--word_repr_size", "64", "--entity_repr_size", "32", "--window_size", "16", "--num_random_entities", "5", "--seed", "1",
                     "--update_method", "adagrad", "--batch_size", "4096", "--nonlinearity", "tanh", "--bias_negative_samples",
                     "--weighting", "uniform", "--max_vocabulary_size", "30000", "--num_epochs", "2", "--output", out, brown])
    assert r.returncode == 0, r.stderr[-3000:]
    assert "corpus size=500" in r.stderr
    costs = epoch_costs(r.stderr)
    assert len(costs) == 2 and costs[1] < costs[0]
    meta = parse_metadata(out + "_meta")
    assert len(meta.object) == 500 and [o.index_object_id for o in meta.object][:3] == [1, 2, 3]
    assert meta.total_terms > 300000


def test_document_list_on_the_indri_repository(tmp_path):
    """--document_list with docnos resolved through the repository's own key files (cpp/data_indri.cpp:693-717): the model's
    documents are the listed ones, in list order."""
    brown = os.path.join(ROOT, "tests", "golden", "Brown_index")
    docnos = ["cj%02d" % i for i in range(80, 0, -1)] + ["ca01", "cr09"]
    lst = tmp_path / "docs.txt"
    lst.write_text("\n".join(docnos) + "\n")
    out = str(tmp_path / "brown_list")
    r = run_trainer(["--word_repr_size", "32", "--entity_repr_size", "16", "--window_size", "8", "--num_random_entities", "3", "--seed", "1",
                     "--update_method", "sgd", "--batch_size", "1024", "--nonlinearity", "tanh", "--weighting", "uniform",
                     "--document_list", str(lst), "--num_epochs", "1", "--output", out, brown])
    assert r.returncode == 0, r.stderr[-3000:]
    assert "corpus size=82" in r.stderr
    meta = parse_metadata(out + "_meta")
    # cj01 … cj80 are documents 295 … 374 of the repository, ca01 is 1, cr09 is 500
    assert [o.index_object_id for o in meta.object] == list(range(374, 294, -1)) + [1, 500]
    assert [o.model_object_id for o in meta.object] == list(range(82))
    bad = tmp_path / "bad.txt"
    bad.write_text("ca01\nnot_a_docno\n")
    r = run_trainer(["--seed", "1", "--update_method", "sgd", "--nonlinearity", "tanh", "--document_list", str(bad), brown])
    assert r.returncode == 1 and "not_a_docno" in r.stderr


def test_multi_gpu_flags_fail_cleanly_on_one_gpu(tmp_path):
    """--gpus N spawns one rank per GPU; on the 1-GPU box it must refuse with a clear message instead of hanging."""
    r = run_trainer(["--gpus", "2", "--seed", "1", "--update_method", "sgd", "--nonlinearity", "tanh", CRANFIELD], timeout=120)
    assert r.returncode == 1 and "--gpus 2 but only 1 HIP device" in r.stderr
    r = run_trainer(["--world_size", "2", "--rank", "2", "--seed", "1", "--update_method", "sgd", "--nonlinearity", "tanh", CRANFIELD], timeout=120)
    assert r.returncode == 1 and "bad --world_size / --rank" in r.stderr
    # --dp_exact_tables is a data-parallel option: a single rank takes it and trains as without it
    r = run_trainer(["--dp_exact_tables", "--seed", "1", "--update_method", "sgd", "--nonlinearity", "tanh", "--num_epochs", "1",
                     "--batch_size", "1024", CRANFIELD], timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_rccl_selftest_and_pinned_alloc():
    import ctypes as C
    import cunvsm_amd as ca
    L = ca.lib()
    ca._lib.check(L.nvsm_comm_selftest(0))
    p = C.c_void_p()
    ca._lib.check(L.nvsm_host_alloc(1 << 20, C.byref(p)))
    assert p.value
    C.memset(p, 7, 1 << 20)
    ca._lib.check(L.nvsm_host_free(p))


def test_initialize_from_rng_state_continues_the_stream():
    """nvsm_rng_set_state + nvsm_initialize_from_rng_state == the reference's model.initialize(&rng) with an RNG that
    was already consumed: identical to seeding directly when nothing was drawn in between."""
    import cunvsm_amd as ca
    from tests.helpers import gpu_model
    spec = dict(num_words=50, num_entities=30, word_dim=8, entity_dim=4, window=3, num_random=2, nonlinearity="tanh",
                batch_norm=False, update_method="sgd")
    spec["lambda"] = 0.0
    a, b = gpu_model(spec, 64), gpu_model(spec, 64)
    a.initialize(17)
    L = ca.lib()
    ca._lib.check(L.nvsm_rng_set_state(b._h, 17))
    ca._lib.check(L.nvsm_initialize_from_rng_state(b._h))
    for name in ("word_representations-representations", "entity_representations-representations", "word_entity_mapping-transform"):
        np.testing.assert_array_equal(a.get_param(name), b.get_param(name))


def _read_dataset(path, name, shape):
    out = path + "." + name + ".bin"
    subprocess.run([H5DUMP, "-d", name, "-b", "LE", "-o", out, path], capture_output=True, check=True)
    return np.fromfile(out, dtype="<f4").reshape(shape)


def _read_batches(path, window):
    raw = open(path, "rb").read()
    off, batches = 0, []
    while off < len(raw):
        n = int(np.frombuffer(raw, "<i8", 1, off)[0]); off += 8
        f = np.frombuffer(raw, "<i8", n * window, off); off += 8 * n * window
        fw = np.frombuffer(raw, "<f4", n * window, off); off += 4 * n * window
        l = np.frombuffer(raw, "<i8", n, off); off += 8 * n
        w = np.frombuffer(raw, "<f4", n, off); off += 4 * n
        batches.append((f, fw, l, w))
    return batches


@pytest.mark.parametrize("method,extra", [("sgd", []), ("sparse_adam", ["--batch_normalization"]), ("adagrad", ["--bias_negative_samples"])])
def test_trainer_trajectory_matches_oracle(tmp_path, method, extra):
    """The whole trainer against the CPU oracle on the same corpus: the data source's batches (dumped by the host-layer
    probe), the shared minstd_rand0 stream (document shuffle → Glorot initialisation → negative sampling), every
    optimiser step of one epoch, and the HDF5 round trip. Initial parameters must agree exactly; the parameters after
    the epoch within the fp32 tolerance of ~20 accumulated steps."""
    from oracle import nvsm_oracle as orc
    from tests.helpers import METHODS
    from tests.test_host_layer import BIN as HOST_TESTS
    if not os.path.exists(HOST_TESTS):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cunvsm_amd", "host"), "build/host_tests"], stdout=subprocess.DEVNULL)
    window, batch, seed, cutoff, dw, de, k, lam, lr = 10, 1024, 3, 200, 48, 32, 5, 0.01, 0.01 if method != "sparse_adam" else 0.001
    nonlin = "hard_tanh" if "--batch_normalization" in extra else "tanh"
    out = str(tmp_path / "m")
    args = ["--word_repr_size", str(dw), "--entity_repr_size", str(de), "--window_size", str(window), "--num_random_entities", str(k),
            "--regularization_lambda", str(lam), "--learning_rate", str(lr), "--weighting", "uniform", "--seed", str(seed),
            "--update_method", method, "--batch_size", str(batch), "--nonlinearity", nonlin, "--max_vocabulary_size", "65536",
            "--min_document_frequency", "0", "--document_cutoff", str(cutoff), "--num_epochs", "1", "--dump_initial_model",
            "--output", out] + extra + [CRANFIELD]
    r = run_trainer(args)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    epoch_bin = str(tmp_path / "epoch.bin")
    info = json.loads(subprocess.run([HOST_TESTS, "--dump-epoch", CRANFIELD, epoch_bin, str(window), str(batch), str(seed), "65536", "0",
                                      str(cutoff)], capture_output=True, text=True, check=True).stdout)
    nV, nD = info["vocabulary"], info["corpus"]
    m, mode = METHODS[method]
    cfg = orc.make_config(nV, nD, dw, de, window, k, batch_norm="--batch_normalization" in extra,
                          nonlinearity=orc.HARD_TANH if nonlin == "hard_tanh" else orc.TANH, clip_sigmoid=True,
                          bias_negative_samples="--bias_negative_samples" in extra, lambda_=lam, update_method=m, adam_mode=mode)
    oracle = orc.Model(cfg, orc.F32)
    rng = orc.Rng(1)
    rng.state = info["rng_state"]
    oracle.initialize(rng)
    shapes = {"word_representations-representations": (nV, dw), "entity_representations-representations": (nD, de),
              "word_entity_mapping-transform": (dw, de), "word_entity_mapping-bias": (1, de)}
    init = {n: _read_dataset(out + "_0.hdf5", n, s) for n, s in shapes.items()}
    for n in shapes:
        np.testing.assert_array_equal(init[n].ravel(), oracle.get(n).astype(np.float32))        # same RNG stream, same fp32 maths
    costs = []
    all_batches = _read_batches(epoch_bin, window)
    for f, fw, l, w in all_batches:
        if len(l) % 1024:
            continue                                                                              # the trainer skips ragged batches
        ids = rng.generate_labels(l, nD, k)
        oracle.forward_native(np.ascontiguousarray(f), np.ascontiguousarray(fw), ids, np.ascontiguousarray(w))
        oracle.backward()
        costs.append(oracle.get_cost())
        oracle.update(lr)
    assert len(costs) >= 10
    logged = epoch_costs(r.stderr)[-1]
    # iterate_data divides the aggregated cost by ALL batches of the epoch, skipped ones included (cpp/main.cu:462,598)
    want = np.sum(costs) / len(all_batches)
    assert abs(logged - want) <= 1e-4 * abs(want), (logged, want)
    final = {n: _read_dataset(out + "_1.hdf5", n, s) for n, s in shapes.items()}
    from tests.helpers import rel_err
    for n in shapes:
        change = np.linalg.norm(oracle.get(n) - init[n].ravel().astype(np.float64))
        diff = np.linalg.norm(final[n].ravel().astype(np.float64) - oracle.get(n))
        # fp32 over ~23 steps: an error relative to the update (Adam's per-component normalisation amplifies it) plus the
        # rounding of `θ·decay + lr·g` itself, which scales with ‖θ‖, not with the update
        tol = (1e-2 if method.endswith("adam") else 2e-3) * change + 2e-6 * np.linalg.norm(init[n])
        assert diff <= tol, (n, diff, change)


@pytest.mark.parametrize("extra", [[], ["--l2_phrase_normalization"], ["--l2_entity_normalization"],
                                   ["--l2_phrase_normalization", "--l2_entity_normalization", "--batch_normalization"]],
                         ids=["plain", "l2_phrase", "l2_entity", "l2_both_bn"])
def test_check_gradients_flag(tmp_path, extra):
    """--check_gradients (cpp/main.cu:414-425 → cpp/gradient_check.cu): central differences for every scalar parameter of
    a small model; passes on the real gradients — also with the optional L2 normalisers, the configurations
    cpp/gradient_checking_tests.cu:92-110 adds — and the run aborts when the check cannot pass (ε far too large)."""
    args = extra + ["--word_repr_size", "3", "--entity_repr_size", "4", "--window_size", "3", "--num_random_entities", "1", "--seed", "2",
            "--update_method", "sgd", "--batch_size", "1024", "--nonlinearity", "tanh", "--weighting", "uniform", "--document_cutoff", "12",
            "--max_vocabulary_size", "30", "--min_document_frequency", "0", "--num_epochs", "1", "--check_gradients", "--allow_ragged_batches",
            "--v", "1", CRANFIELD]
    r = run_trainer(args, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert re.search(r"Gradient check: \d+ parameters, passed", r.stderr)
    assert "incorrect direction" not in r.stderr
