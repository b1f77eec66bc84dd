"""Shared helpers of the parity tests: build the same model on the CPU oracle and on the HIP path."""
import numpy as np

from oracle import nvsm_oracle as orc

PARAMS = (
    "word_representations-representations",
    "entity_representations-representations",
    "word_entity_mapping-transform",
    "word_entity_mapping-bias",
)

METHODS = {
    "sgd": (orc.SGD, orc.ADAM_NONE),
    "adagrad": (orc.ADAGRAD, orc.ADAM_NONE),
    "sparse_adam": (orc.ADAM, orc.ADAM_SPARSE),
    "dense_adam": (orc.ADAM, orc.ADAM_DENSE_UPDATE),
    "full_adam": (orc.ADAM, orc.ADAM_DENSE_UPDATE_DENSE_VARIANCE),
}


def oracle_model(spec, dtype=orc.F64):
    method, mode = METHODS[spec.get("update_method", "sgd")]
    cfg = orc.make_config(spec["num_words"], spec["num_entities"], spec["word_dim"], spec["entity_dim"], spec["window"],
                          spec["num_random"], batch_norm=spec.get("batch_norm", False),
                          nonlinearity=orc.HARD_TANH if spec.get("nonlinearity", "tanh") == "hard_tanh" else orc.TANH,
                          clip_sigmoid=spec.get("clip_sigmoid", True),
                          bias_negative_samples=spec.get("bias_negative_samples", False),
                          l2_phrase=spec.get("l2_phrase", False), l2_entity=spec.get("l2_entity", False),
                          lambda_=spec.get("lambda", 0.0), update_method=method, adam_mode=mode)
    # the OpenMP team: every core for the full-size problems, a handful for the small ones (a parallel region over the
    # 256 hardware threads of the GPU box costs more than a small loop)
    big = max(spec["num_words"] * spec["word_dim"], spec["num_entities"] * spec["entity_dim"]) >= 4_000_000
    orc.set_num_threads(0 if big else 8)
    return orc.Model(cfg, dtype)


def gpu_model(spec, max_batch, **extra):
    import cunvsm_amd as ca
    cfg = ca.default_config(num_words=spec["num_words"], num_entities=spec["num_entities"],
                            word_repr_size=spec["word_dim"], entity_repr_size=spec["entity_dim"],
                            window_size=spec["window"], num_random_entities=spec["num_random"],
                            batch_normalization=spec.get("batch_norm", False),
                            nonlinearity=spec.get("nonlinearity", "tanh"),
                            clip_sigmoid=spec.get("clip_sigmoid", True),
                            bias_negative_samples=spec.get("bias_negative_samples", False),
                            l2_normalize_phrase_reprs=int(spec.get("l2_phrase", False)),
                            l2_normalize_entity_reprs=int(spec.get("l2_entity", False)),
                            regularization_lambda=spec.get("lambda", 0.0),
                            update_method=spec.get("update_method", "sgd"), max_batch_size=max_batch, **extra)
    return ca.Model(cfg)


def random_params(spec, rs, scale=None):
    """Glorot-scaled random parameters in the reference's raw layouts."""
    nV, nD, dw, de = spec["num_words"], spec["num_entities"], spec["word_dim"], spec["entity_dim"]
    g = lambda r, c: np.sqrt(6.0 / (r + c))
    out = {
        PARAMS[0]: rs.uniform(-1, 1, nV * dw) * (scale or g(dw, nV)),
        PARAMS[1]: rs.uniform(-1, 1, nD * de) * (scale or g(de, nD)),
        PARAMS[2]: rs.uniform(-1, 1, de * dw) * (scale or g(de, dw)),
        PARAMS[3]: rs.uniform(-0.1, 0.1, de),
    }
    return {k: v.astype(np.float32) for k, v in out.items()}


def load_params(model, params, is_gpu):
    for k, v in params.items():
        if is_gpu:
            model.set_param(k, v)
        else:
            model.set(k, v.astype(np.float64))


def zipf_ids(rs, n, size, s=1.0):
    p = 1.0 / np.arange(1, n + 1) ** s
    p /= p.sum()
    return rs.choice(n, size=size, p=p).astype(np.int64)


def random_batch(spec, rs, B, zipf=False, weighted=True):
    nV, nD, w, k = spec["num_words"], spec["num_entities"], spec["window"], spec["num_random"]
    words = zipf_ids(rs, nV, B * w) if zipf else rs.randint(0, nV, B * w).astype(np.int64)
    ww = rs.uniform(0.0, 2.0, B * w).astype(np.float32) if weighted else np.ones(B * w, np.float32)
    labels = rs.randint(0, nD, B).astype(np.int64)
    iw = rs.uniform(0.0, 2.0, B).astype(np.float32) if weighted else np.ones(B, np.float32)
    ids = rs.randint(0, nD, (B, k + 1)).astype(np.int64)
    ids[:, 0] = labels
    return words, ww, labels, iw, ids.ravel()


def rel_err(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
