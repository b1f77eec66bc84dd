#!/usr/bin/env python3
"""Harvests the numeric golden vectors the reference's own googletest files hold for the hot path and
writes them as JSON fixtures next to this script. Run once in the build container (where
/root/reference exists); the JSON files are committed, /root/reference is never read at test time.

Only literal numbers (inputs / expected outputs) are extracted — no reference source text is kept.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def lines(path, lo, hi):
    with open(os.path.join(REF, path)) as f:
        src = f.read().split("\n")
    return "\n".join(src[lo - 1:hi])


def eq_literals(text):
    """numbers wrapped as FPHelper<FloatT>::eq(<literal>)"""
    return [float(x) for x in re.findall(r"eq\(\s*(" + NUM + r")\s*\)", text)]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


# --- cpp/model_tests.cu:341-466  ParamsTest.Transform_backward --------------------------------
t = "cpp/model_tests.cu"
dump("transform_backward.json", {
    "source": "cpp/model_tests.cu:341-466 (ParamsTest.Transform_backward)",
    "seed": 10, "num_words": 5, "num_entities": 3, "word_dim": 2, "entity_dim": 3,
    "bias_negative_samples": True, "num_random_entities": 10, "regularization_lambda": 0.01,
    "update_method": "SGD", "batch_size": 32, "window_size": 2,
    "feature_value": 2, "label": 1, "feature_weight": 1.0, "instance_weight": 1.0,
    "nonlinearity": "TANH", "batch_normalization": False, "clip_sigmoid": False,
    "grad_transform": eq_literals(lines(t, 377, 386)),
    "grad_bias": eq_literals(lines(t, 389, 395)),
    "grad_phrase": eq_literals(lines(t, 398, 465)),
})

# --- cpp/model_tests.cu:468-521  ParamsTest.Transform_BatchNormalization ------------------------
dump("transform_batchnorm.json", {
    "source": "cpp/model_tests.cu:468-548 (ParamsTest.Transform_BatchNormalization)",
    "word_dim": 3, "entity_dim": 5, "bn_epsilon": 1e-5, "nonlinearity": "TANH",
    "transform": list(range(15)), "bias": [i * 1e-3 for i in range(5)],
    "input": [0.01, 0.02, 0.03, 0.001, 0.002, 0.003],
    "output": eq_literals(lines(t, 509, 521)),
    "grad_output": [0.1] * 5 + [0.2] * 5,
})

# --- cpp/cudnn_utils_tests.cu:114-176  BatchNormalization_forward_backward ---------------------
t = "cpp/cudnn_utils_tests.cu"
dump("batchnorm_forward_backward.json", {
    "source": "cpp/cudnn_utils_tests.cu:114-176 (cuDNNTests.BatchNormalization_forward_backward)",
    "num_features": 3, "num_instances": 2, "epsilon": 1e-5,
    "input": [1.0, 2.0, 3.0, 5.0, 10.0, 20.0],
    "mean": [3.0, 6.0, 11.5], "variance": [4.0, 16.0, 72.25],
    "grad": [0.25, -0.1, 0.3, 1.0, 0.005, -0.5],
    "grad_bias": eq_literals(lines(t, 163, 167)),
    "grad_input": eq_literals(lines(t, 169, 176)),
})

# --- cpp/cuda_utils_tests.cu:51-92  CudaUtilsTest.Normalizer -----------------------------------
t = "cpp/cuda_utils_tests.cu"
block = lines(t, 81, 92)
dump("normalizer.json", {
    "source": "cpp/cuda_utils_tests.cu:51-92 (CudaUtilsTest.Normalizer)",
    "num_features": 5, "num_instances": 2,
    "input": list(range(1, 11)),
    "grad_output": list(range(10000, 10010)),
    "grad_input": [float(x) for x in re.findall(r"(?:^|\()\s*(" + NUM + r")[,)]", block, flags=re.M)],
})

# --- cpp/cuda_utils_tests.cu:8-21  truncated_sigmoid -------------------------------------------
dump("truncated_sigmoid.json", {
    "source": "cpp/cuda_utils_tests.cu:8-21 (CudaFuncTests.truncated_sigmoid)",
    "sigmoid_1": eq_literals(lines(t, 12, 12))[0],
    "cases": [
        {"x": 0.0, "eps": 0.0, "expect": 0.5},
        {"x": -100.0, "eps": 1e-7, "expect": 1e-7},
        {"x": 100.0, "eps": 1e-7, "expect": 1.0 - 1e-7},
    ],
})

# --- cpp/updates_tests.cu:299-425  AdamTransformGradientUpdater (literal bias goldens) ---------
t = "cpp/updates_tests.cu"
dump("adam_transform.json", {
    "source": "cpp/updates_tests.cu:299-425 (UpdatesTest.AdamTransformGradientUpdater)",
    "epsilon": 1e-5, "beta1": 0.9, "beta2": 0.999, "initial": 5.0, "word_dim": 8, "entity_dim": 3,
    "grad_matrix": [float(i) for i in range(1, 25)], "grad_bias": [25.0, 26.0, 27.0],
    "step1": {"grad_bias_out": eq_literals(lines(t, 350, 354)),
              "m_bias": eq_literals(lines(t, 356, 360)),
              "v_bias": eq_literals(lines(t, 362, 366))},
    "step2": {"grad_bias_out": eq_literals(lines(t, 407, 411)),
              "m_bias": eq_literals(lines(t, 413, 417)),
              "v_bias": eq_literals(lines(t, 419, 423))},
    "params": [[0.0, 1.0], [0.0, 0.5], [0.1, 1.0], [0.1, 0.5]],
})

# --- cpp/updates_tests.cu:250-297  Adagrad row-scalar accumulator literals ----------------------
dump("adagrad_representations.json", {
    "source": "cpp/updates_tests.cu:250-297 (UpdatesTest.AdagradRepresentationsGradientUpdater)",
    "epsilon": 1e-6, "initial": 5.0, "num_objects": 10, "repr_size": 4, "window_size": 3,
    "grad": [2.0, 2.5, 3.0, 4.0, 10.0, 11.0, 12.0, 13.0], "indices": [9, 0, 1, 5, 1, 8],
    "accumulator": [float(x) for x in re.findall(NUM, lines(t, 285, 285))],
})
