"""CPU tests of the host layer above the C ABI (cunvsm_amd/host: batch container, data sources, IndriSource logic
over an abstract index, TREC-text index, async prefetch, Metadata wire format, HDF5 writer, flag parser).

tests/cpp/host_tests.cpp restates the reference's own googletest cases (cpp/data_tests.cpp, cpp/utils_tests.cpp)
with the same inputs and expected values — including the seed-pinned instance order of
IndriSourceTest.StochasticIndriSource_SelfInformation — and is built here with g++; this file checks the verdict of
every case and cross-examines the files the host layer writes with independent readers (python-protobuf for
"<output>_meta", h5dump for the checkpoint)."""
import json
import os
import re
import shutil
import subprocess

import pytest

from tests.conftest import ROOT

HOST_DIR = os.path.join(ROOT, "cunvsm_amd", "host")
BIN = os.path.join(HOST_DIR, "build", "host_tests")
CRANFIELD = os.path.join(ROOT, "tests", "golden", "cranfield", "cranfield.trectext")
BROWN = os.path.join(ROOT, "tests", "golden", "Brown_index")       # the Indri 5.8 repository of the reference's own test
os.environ["NVSM_BROWN_INDEX"] = BROWN

CASES = [
    "InMemoryDocumentSource.InMemoryDocumentSource", "InMemoryDocumentSource.pad_batch",
    "DataSourceTest.create_instances", "DataSourceTest.create_instances_overflow",
    "IndriSourceTest.IndriSource", "IndriSourceTest.IndriSource_UnsupportedSampling_Death",
    "IndriSourceTest.StochasticIndriSource", "IndriSourceTest.StochasticIndriSource_Resampling",
    "IndriSourceTest.StochasticIndriSource_SelfInformation",
    "MetaSourceTest.AsyncSource", "MetaSourceTest.RepeatingSource",
    "Base.utils", "Batch.swap", "Metadata.roundtrip", "TrectextIndex.end_to_end", "IndriSourceTest.Brown", "IndriRepository.docno_lookups",
    "DataParallel.rendezvous_file",
]


@pytest.fixture(scope="module")
def host_tests():
    subprocess.check_call(["make", "-C", HOST_DIR, "build/host_tests"], stdout=subprocess.DEVNULL)
    return BIN


@pytest.mark.parametrize("case", CASES)
def test_reference_data_test_case(host_tests, case):
    r = subprocess.run([host_tests, case], capture_output=True, text=True, timeout=120)
    assert "[PASS] %s" % case in r.stdout, r.stdout + r.stderr
    assert "skipped" not in r.stdout
    assert r.returncode == 0


def test_every_case_is_listed(host_tests):
    r = subprocess.run([host_tests], capture_output=True, text=True, timeout=300)
    ran = re.findall(r"\[(?:PASS|FAIL)\] (\S+)", r.stdout)
    assert sorted(ran) == sorted(CASES)
    assert r.stdout.strip().endswith("0 failed")


def _metadata_class():
    """lse.Metadata (proto/nvsm.proto:88-103) declared at run time with python-protobuf: an independent reader."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="nvsm_meta_test.proto", package="lse_test", syntax="proto3")
    m = fd.message_type.add(name="Metadata")
    t = m.nested_type.add(name="TermInfo")
    for i, n in enumerate(["index_term_id", "model_term_id", "term_frequency"], 1):
        t.field.add(name=n, number=i, type=descriptor_pb2.FieldDescriptorProto.TYPE_INT32, label=descriptor_pb2.FieldDescriptorProto.LABEL_OPTIONAL)
    o = m.nested_type.add(name="ObjectInfo")
    for i, n in enumerate(["index_object_id", "model_object_id"], 1):
        o.field.add(name=n, number=i, type=descriptor_pb2.FieldDescriptorProto.TYPE_INT32, label=descriptor_pb2.FieldDescriptorProto.LABEL_OPTIONAL)
    m.field.add(name="term", number=1, type=descriptor_pb2.FieldDescriptorProto.TYPE_MESSAGE, type_name=".lse_test.Metadata.TermInfo",
                label=descriptor_pb2.FieldDescriptorProto.LABEL_REPEATED)
    m.field.add(name="object", number=2, type=descriptor_pb2.FieldDescriptorProto.TYPE_MESSAGE, type_name=".lse_test.Metadata.ObjectInfo",
                label=descriptor_pb2.FieldDescriptorProto.LABEL_REPEATED)
    m.field.add(name="total_terms", number=3, type=descriptor_pb2.FieldDescriptorProto.TYPE_INT32, label=descriptor_pb2.FieldDescriptorProto.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    desc = pool.FindMessageTypeByName("lse_test.Metadata")
    if hasattr(message_factory, "GetMessageClass"):
        return message_factory.GetMessageClass(desc)
    return message_factory.MessageFactory(pool).GetPrototype(desc)


def parse_metadata(path):
    msg = _metadata_class()()
    with open(path, "rb") as f:
        msg.ParseFromString(f.read())
    return msg


def test_metadata_wire_format_is_protobuf(host_tests, tmp_path):
    path = str(tmp_path / "model_meta")
    subprocess.check_call([host_tests, "--write-meta", path])
    msg = parse_metadata(path)
    assert [(t.index_term_id, t.model_term_id, t.term_frequency) for t in msg.term] == \
        [(100 + i, i, 7 * i + 1) for i in range(5)] + [(-3, 5, 2147483647)]
    assert [(o.index_object_id, o.model_object_id) for o in msg.object] == [(1, 0), (2, 1), (3, 2)]
    assert msg.total_terms == 260760
    # and protobuf's own serialisation of the same message is byte-identical (field order, default elision, varints)
    with open(path, "rb") as f:
        assert msg.SerializeToString() == f.read()


H5DUMP = shutil.which("h5dump") or "/opt/conda/bin/h5dump"


def h5_header(path):
    out = subprocess.run([H5DUMP, "-H", path], capture_output=True, text=True, check=True).stdout
    return {n: (dt, tuple(int(x) for x in dims.split(","))) for n, dt, dims in
            re.findall(r'DATASET "([^"]+)" \{\s*DATATYPE\s+(\S+)\s*DATASPACE\s+SIMPLE \{ \(([^)]*)\)', out)}


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="h5dump not available")
def test_hdf5_checkpoint_layout(host_tests, tmp_path):
    """Dataset names, shapes ([objects][dim], [word_dim][entity_dim], [1][entity_dim]), float32 LE and H5F_ACC_EXCL:
    cpp/hdf5.cu:26-53, lse_hdf5_inl.h:20-26, py/nvsm/base.py:182-240."""
    path = str(tmp_path / "model_1.hdf5")
    subprocess.check_call([host_tests, "--write-hdf5", path])
    ds = h5_header(path)
    assert ds == {"entity_representations-representations": ("H5T_IEEE_F32LE", (4, 2)),
                  "word_entity_mapping-bias": ("H5T_IEEE_F32LE", (1, 2)),
                  "word_entity_mapping-transform": ("H5T_IEEE_F32LE", (3, 2)),
                  "word_representations-representations": ("H5T_IEEE_F32LE", (5, 3))}
    out = subprocess.run([H5DUMP, "-d", "word_entity_mapping-bias", path], capture_output=True, text=True, check=True).stdout
    assert "0.25, -0.75" in out
    # the reference opens with H5F_ACC_EXCL: an existing file is an error, not an overwrite
    r = subprocess.run([host_tests, "--write-hdf5", path], capture_output=True, text=True)
    assert r.returncode != 0


def test_cranfield_collection_statistics(host_tests):
    """The TREC-text index against the figures Indri reports for the same file (TUTORIAL.md:34-43: 1400 documents,
    260 760 occurrences, lengths 3..698); the token count differs by 0.12 % (acronym / apostrophe folding)."""
    r = subprocess.run([host_tests, "--index-stats", CRANFIELD], capture_output=True, text=True, check=True)
    st = json.loads(r.stdout)
    assert st["documents"] == 1400 and st["min_len"] == 3 and st["max_len"] == 698
    assert abs(st["tokens"] - 260760) / 260760 < 0.002
    assert st["corpus"] == 1398                       # two abstracts are shorter than the 10-word window
    assert st["batches"] == -(-st["instances"] // 4096)
    # seed 1 must give the same shuffled stream on every run and machine (minstd_rand0 + fixed shuffle algorithm)
    assert st["feature_checksum"] == 224620507 and st["instances"] == 150984 and st["vocabulary"] == 4897


def test_trainer_flag_surface():
    """Every option of the reference CLI (cpp/main.cu:15-76) is defined with the same name and default."""
    src = open(os.path.join(HOST_DIR, "train_main.cpp")).read()
    ref = {"num_epochs": "100000", "document_cutoff": "0", "document_list": '""', "term_blacklist": '""', "word_repr_size": "4",
           "entity_repr_size": "4", "batch_size": "1024", "window_size": "8", "num_random_entities": "1", "seed": "0",
           "regularization_lambda": "0.01", "learning_rate": "0.0", "update_method": '""', "weighting": '"auto"',
           "feature_weighting": '"uniform"', "bias_negative_samples": "false", "nonlinearity": '""',
           "l2_phrase_normalization": "false", "l2_entity_normalization": "false", "batch_normalization": "false",
           "max_vocabulary_size": "60000", "min_document_frequency": "2", "max_document_frequency": "0.5", "include_oov": "false",
           "compute_initial_cost": "false", "check_gradients": "false", "no_shuffle": "false", "dump_initial_model": "false",
           "dump_every": "0", "entity_similarity_weight": "0.0", "term_similarity_weight": "0.0", "output": '""'}
    for name, default in ref.items():
        m = re.search(r'define_\w+\("%s", &FLAGS_%s, ([^,]+),' % (name, name), src)
        assert m, name
        assert m.group(1).strip() == default, (name, m.group(1))
