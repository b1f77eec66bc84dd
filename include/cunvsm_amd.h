/*
 * cunvsm_amd — C ABI of the MI355X-native NVSM / LSE training hot path.
 *
 * Drop-in boundary for cuNVSM's `Model<TextEntity::Objective>` (include/cuNVSM/model.h:75-131):
 * the per-batch compute_cost → compute_gradients → update → get_cost sequence driven by
 * iterate_data (cpp/main.cu:400-444) and ModelTest::train (include/cuNVSM/tests_base_cuda.h:161-190).
 * Plain pointers and sizes only; the library owns every device-side object, the caller owns its
 * host buffers; nothing but the opaque handle crosses the boundary. Every entry point returns an
 * nvsm_status (0 = OK) instead of aborting (the reference CHECK()s / LOG(FATAL)s).
 *
 * Layouts are the reference's raw buffers:
 *   embedding tables   [num_objects][dim] row-major  (device_matrix dim x n, column-major)
 *   projection         entity_dim x word_dim column-major: T[r + entity_dim * c]
 *   bias               [entity_dim]
 *   indices            int64 (include/cuNVSM/base.h:28  `typedef long int32`)
 *   floating point     float32 (release build, cpp/CMakeLists.txt:17)
 */
#ifndef CUNVSM_AMD_H
#define CUNVSM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nvsm_model nvsm_model;

typedef enum {
    NVSM_OK = 0,
    NVSM_ERR_INVALID_ARGUMENT = 1,
    NVSM_ERR_UNSUPPORTED = 2,      /* configuration outside the kernels' limits (entity_repr_size > 1024, the reference's own limit) */
    NVSM_ERR_DEVICE = 3,           /* HIP / RCCL error, see nvsm_last_error() */
    NVSM_ERR_STATE = 4,            /* call sequence violated (e.g. compute_gradients before compute_cost) */
    NVSM_ERR_NO_DEVICE = 5         /* no MI355X visible: there is NO CPU fallback */
} nvsm_status;

/* proto/nvsm.proto:11-14 (ModelDesc.TransformDesc.Nonlinearity) */
enum { NVSM_TANH = 0, NVSM_HARD_TANH = 1 };
/* proto/nvsm.proto:40-44 (TrainConfig.UpdateMethod) */
enum { NVSM_SGD = 0, NVSM_ADAGRAD = 1, NVSM_ADAM = 2 };
/* proto/nvsm.proto:50-55 (AdamConf.AdamMode); NONE behaves as SPARSE (cpp/updates_adam.cu:289,332) */
enum { NVSM_ADAM_NONE = 0, NVSM_ADAM_SPARSE = 1, NVSM_ADAM_DENSE_UPDATE = 2, NVSM_ADAM_DENSE_UPDATE_DENSE_VARIANCE = 3 };
/* negative sampling: F2, cpp/objective.cu:5-28 → cpp/labels.cu:4-22 */
enum {
    NVSM_SAMPLER_HOST_MINSTD = 0,  /* std::minstd_rand0 + fresh uniform_int_distribution<long> per draw: draw-for-draw the reference */
    NVSM_SAMPLER_DEVICE = 1        /* counter-based hash on the GPU, same distribution (uniform over all documents) */
};

/*
 * Replaces the constructor arguments Model(num_words, num_entities, ModelDesc, TrainConfig)
 * (cpp/model.cu:95-103); field names follow proto/nvsm.proto:7-71.
 */
typedef struct {
    int64_t num_words;
    int64_t num_entities;
    /* ModelDesc */
    int32_t word_repr_size;
    int32_t entity_repr_size;
    int32_t batch_normalization;       /* transform_desc.batch_normalization */
    int32_t nonlinearity;              /* transform_desc.nonlinearity */
    int32_t clip_sigmoid;              /* forced true by the reference CLI (cpp/main.cu:645) */
    int32_t bias_negative_samples;
    int32_t l2_normalize_phrase_reprs; /* objective.cu:99-103,136-142,461-468: optional; separate normaliser passes */
    int32_t l2_normalize_entity_reprs; /* objective.cu:104-107,168-174,405-412: optional; generic loss kernel, gradient rows materialised */
    /* TrainConfig */
    int32_t window_size;
    int32_t num_random_entities;
    float   regularization_lambda;
    int32_t update_method;
    int32_t adam_mode;
    int32_t max_batch_size;            /* TrainConfig.batch_size: capacity of the per-step device buffers (per rank) */
    /* runtime */
    int32_t device;                    /* HIP device ordinal */
    int32_t sampler;                   /* NVSM_SAMPLER_* */
    /* data parallelism (new). world_size 1 = single GPU. */
    int32_t world_size;
    int32_t rank;
    int32_t sync_batch_norm;           /* 1: global batch statistics (exact single-GPU maths); 0: per-shard */
    int32_t dp_exact_tables;           /* world_size > 1 — 0: embedding tables are updated from the rank's own windows (replicas drift);
                                          1: every rank applies the sparse gradients of ALL ranks' windows (all-gather of the update's
                                          inputs in front of the table passes): replicas stay bit-identical and follow the single-GPU
                                          trajectory on the global batch, at world_size x the table-update work per rank (see below) */
    int32_t reserved[4];
} nvsm_config;

/* Fills the reference CLI defaults (cpp/main.cu:15-76,637-721; scripts/functions.sh:380-399). */
void nvsm_config_default(nvsm_config* cfg);

/*
 * Replaces TextEntity::Batch (include/cuNVSM/data.h:114-177, cpp/data.cu:8-124): four flat arrays.
 * on_device = 0: host pointers (the reference's pinned-host batch). Page-locked arrays (nvsm_host_alloc, hipHostMalloc,
 *                hipHostRegister) are read over PCIe by a kernel on the engine's copy stream, a step ahead of their use, and
 *                the call never holds the calling thread; pageable arrays are staged by hipMemcpyAsync, which does;
 * on_device = 1: device pointers already resident in HBM (no copy).
 */
typedef struct {
    const int64_t* features;          /* [num_instances * window_size] word ids */
    const float*   feature_weights;   /* [num_instances * window_size] or NULL (= all 1.0) */
    const int64_t* labels;            /* [num_instances] document ids */
    const float*   weights;           /* [num_instances] or NULL (= all 1.0) */
    int64_t        num_instances;
    int32_t        on_device;
} nvsm_batch;

const char* nvsm_last_error(void);
const char* nvsm_version(void);
/* number of visible HIP devices (0 ⇒ nvsm_create fails with NVSM_ERR_NO_DEVICE) */
int nvsm_device_count(void);

/* Bind the CALLING host thread (and the threads it creates afterwards) to the CPUs of the NUMA node the device hangs off
 * (/sys/bus/pci/devices/<bus id>/local_cpulist, intersected with the thread's present affinity mask; left as it is when the
 * intersection is empty or the node is unknown). The steps of a small batch are a chain of ~45 launches and event calls per
 * 0.15 ms: queued from the far socket they take the host LONGER than they take the GPU (LSE recipe, batch 4096, two-socket
 * host: 0.159 ms per step unbound or on the far node, 0.150 on the device's node). The reference has no counterpart (one
 * GPU, one thread, cpp/main.cu:623-767); the trainer and bench.py call it once per process, right after choosing the device.
 * Call it BEFORE the first HIP call of the process where possible: the device is then found through sysfs (the AMD render nodes the
 * process may open, in order; nvsm_create re-checks the guess against hipDeviceGetPCIBusId and re-binds if it was wrong), and the
 * runtime's own threads and host allocations land on the device's node as well (LSE: 0.1495 ms in every process against
 * 0.150-0.162 when bound after the runtime came up). Where sysfs does not identify the device the runtime is asked (and comes up).
 * *numa_node (may be null): the device's node, -1 if unknown. NVSM_BIND_HOST=0 in the environment makes it a no-op. */
int nvsm_bind_host_thread(int device, int* numa_node);

/* Model::Model (cpp/model.cu:95-103) / ~Model */
int nvsm_create(const nvsm_config* cfg, nvsm_model** out);
void nvsm_destroy(nvsm_model* m);

/* ModelBase::initialize(RNG*) (cpp/model.cu:37-43): Glorot-uniform words → entities → transform, bias = 0,
 * drawn from std::minstd_rand0(seed) exactly as include/cuNVSM/cuda_utils.h:35-56; the same generator then
 * feeds the host negative sampler (as `rng` does in cpp/main.cu:729-730,405). */
int nvsm_initialize(nvsm_model* m, uint64_t seed);
int nvsm_rng_get_state(nvsm_model* m, uint64_t* state);   /* `rng_state << *rng`  (cpp/main.cu:401-402) */
int nvsm_rng_set_state(nvsm_model* m, uint64_t state);
/* model.initialize(&rng) with a generator that has ALREADY been consumed: the reference seeds one RNG in main()
 * (cpp/main.cu:729-730), lets the data source draw from it first (document shuffling, cpp/main.cu:497-499) and only
 * then initialises the parameters (cpp/main.cu:520). The caller installs that state with nvsm_rng_set_state and
 * calls this instead of nvsm_initialize(seed). */
int nvsm_initialize_from_rng_state(nvsm_model* m);

/* Pinned host memory for batches handed over with on_device = 0 — what TextEntity::Batch allocates with
 * cudaHostAlloc (cpp/data.cu:16-27), so that bringing a batch into HBM is truly asynchronous (see nvsm_batch). */
int nvsm_host_alloc(size_t bytes, void** out);
int nvsm_host_free(void* p);

/* ModelBase::get_data() (cpp/model.cu:64-93) — the four tensors write_to_hdf5 dumps. Names:
 *   "word_representations-representations"   [num_words][word_repr_size]
 *   "entity_representations-representations" [num_entities][entity_repr_size]
 *   "word_entity_mapping-transform"          entity_repr_size x word_repr_size, column-major
 *   "word_entity_mapping-bias"               [entity_repr_size]
 * nvsm_set_param is the test hook that replaces Storage::increment_parameter / initialize_with_constant
 * (cpp/storage.cu:108-131,252-264). Optimiser state is reachable under "<param>/m", "<param>/v", "<param>/a". */
int nvsm_param_size(nvsm_model* m, const char* name, int64_t* count);
int nvsm_get_param(nvsm_model* m, const char* name, float* host_dst, int64_t count);
int nvsm_set_param(nvsm_model* m, const char* name, const float* host_src, int64_t count);
/* Storage::increment_parameter(idx, epsilon) (cpp/storage.cu:123-131,252-264): the gradient checker's poke. */
int nvsm_increment_parameter(nvsm_model* m, const char* name, int64_t index, float delta);

/* Index contract (the reference indexes its tables with these ids unchecked, cpp/params.cu:75-95, cpp/storage.cu:37-49):
 * every word id must be in [0, num_words), every label / entity id in [0, num_entities). The arrays must hold
 * num_instances * window_size (features, feature_weights), num_instances (labels, weights) and
 * num_instances * (num_random_entities + 1) (entity_ids) elements. Ids are checked ON THE DEVICE when they are narrowed
 * to 32 bits: an id out of range is replaced by row 0 (nothing is ever read or written out of bounds) and the NEXT
 * host-side wait of the handle — nvsm_get_cost, nvsm_deferred_cost, nvsm_synchronize, nvsm_get_param / nvsm_get_tensor —
 * returns NVSM_ERR_INVALID_ARGUMENT once; the numbers of that step are meaningless. With the environment variable
 * NVSM_DEBUG=1 (the reference's debug build: CHECK_MATRIX, cpp/objective.cu:134,152) compute_cost / compute_gradients
 * additionally verify that every intermediate and parameter is finite and report at once (NVSM_ERR_DEVICE). */
/* Model::compute_cost(batch, rng) (cpp/model.cu:135-143 → cpp/objective.cu:30-313).
 * entity_ids: optional [num_instances * (num_random_entities + 1)] int64 HOST array laid out as
 * generate_labels does ([label, neg_1..neg_k] per instance); NULL ⇒ sampled per cfg.sampler. */
int nvsm_compute_cost(nvsm_model* m, const nvsm_batch* batch, const int64_t* entity_ids);
/* Model::compute_gradients(result) (cpp/model.cu:145-152 → cpp/objective.cu:315-481) */
int nvsm_compute_gradients(nvsm_model* m);
/* Model::update(gradients, learning_rate, scaled_regularization_lambda) (cpp/model.cu:187-220) */
int nvsm_update(nvsm_model* m, float learning_rate, float scaled_regularization_lambda);
/* ForwardResult::get_cost() (cpp/intermediate_results.cu:80-124) — synchronises the stream, like the reference. */
int nvsm_get_cost(nvsm_model* m, float* cost);
/* the same value before it is narrowed to FloatT: the device accumulates Σ ω·log p in fp64, which is what lets the
 * gradient checker difference two costs that agree to seven digits */
int nvsm_get_cost_f64(nvsm_model* m, double* cost);
/* ForwardResult::scaled_regularization_lambda() (cpp/intermediate_results.cu:126-129): lambda / (global) batch */
float nvsm_scaled_regularization_lambda(nvsm_model* m);

/* One iterate_data loop body (cpp/main.cu:400-444): compute_cost + compute_gradients + update with
 * scaled lambda; fully asynchronous. cost may be NULL (no read-back, no sync). With a cost pointer the call returns once
 * the step's loss kernel has run and its loss word has been copied out (single GPU: the backward pass and the updates
 * are still running then, and the caller can queue the next step); under data parallelism, where the word is summed
 * over the ranks in the backward pass, it waits for the step as nvsm_get_cost does.
 * LIFETIME of a device-resident batch (batch->on_device): the word update reads batch->feature_weights — and the loss
 * kernel batch->weights — long after the prologue has consumed the ids, and nvsm_step(cost) returns BEFORE the updates have
 * run: the caller must leave all four arrays untouched until nvsm_synchronize — or, when the handle runs on the caller's own
 * stream (nvsm_set_stream), order the refill on that stream behind the step (everything that reads the batch is on that stream
 * or joined to it by the next call on the handle). Double-buffer device batches otherwise. nvsm_wait_inputs covers host batches
 * only. Errors flagged by the backward and update kernels of a step surface at the next call that waits. */
int nvsm_step(nvsm_model* m, const nvsm_batch* batch, const int64_t* entity_ids, float learning_rate, float* cost);

/* The same step for training loops that want the loss of EVERY batch, as cpp/main.cu:427-444 does, without putting the
 * GPU behind the host: nvsm_step_deferred queues the step plus a device→host copy of its loss word and hands back a
 * ticket; nvsm_deferred_cost(ticket) waits for that copy only (not for the step's updates). At most
 * NVSM_MAX_DEFERRED tickets may be outstanding (older ones are overwritten). nvsm_wait_inputs returns once the last
 * queued step has copied its host batch to the device, i.e. once the caller may refill those host buffers; a loop that
 * calls it before refilling and reads each loss one step late keeps one whole step queued ahead of the GPU. */
#define NVSM_MAX_DEFERRED 8
int nvsm_step_deferred(nvsm_model* m, const nvsm_batch* batch, const int64_t* entity_ids, float learning_rate, int64_t* ticket);
int nvsm_deferred_cost(nvsm_model* m, int64_t ticket, float* cost);
int nvsm_wait_inputs(nvsm_model* m);

/* Intermediates / gradients of the last compute_cost / compute_gradients, for parity tests and the
 * gradient checker (Parameters::get_parameter_gradient, cpp/storage.cu:133-183,264-283). Names:
 *   "phrase" [B][dw], "pre" [B][de], "proj" [B][de], "probs" [B*R], "entity_ids" [B*R] (as float),
 *   "bn_mean" [de], "bn_inv_std" [de], "multipliers" [B*R] (signed),
 *   "grad_transform" (de x dw col-major), "grad_bias" [de], "grad_phrase" [B][dw], "grad_entity" [B*R][de],
 *   "grad_proj" [B][de]. */
int nvsm_tensor_size(nvsm_model* m, const char* name, int64_t* count);
int nvsm_get_tensor(nvsm_model* m, const char* name, float* host_dst, int64_t count);

/* Streams. A handle issues its work on FOUR HIP streams of its own device: the main stream (highest priority: the step's
 * critical chain), two side streams (lowest priority: the batch → row-order sorts, and — in nvsm_step — the documents
 * update and the ∂T GEMM + projection update, which keep running after nvsm_step has returned and are joined by the next
 * step where it needs their results) and a copy stream (host batches → HBM). The reference collapses to one stream,
 * cpp/model.cu:13-14. nvsm_set_stream replaces the MAIN stream only (the caller's stream keeps its own priority; the
 * side streams still order themselves against it with events); NULL = a fresh highest-priority stream of the handle's
 * own. nvsm_synchronize waits for all four and reports any error a kernel has flagged since the last wait. */
int nvsm_set_stream(nvsm_model* m, void* hip_stream);
int nvsm_synchronize(nvsm_model* m);
/* One line of text: which kernel each of a step's three projection products takes at `batch` windows on this handle, where the
 * dT product and the CSR builds run, whether the tables decay lazily, and every NVSM_* switch that is off its default (the
 * switches are read from the environment ONCE, by nvsm_create: INTEGRATION.md §6). No reference counterpart. */
int nvsm_describe(nvsm_model* m, int64_t batch, char* buf, int64_t buf_bytes);

/* Data parallelism over RCCL / xGMI: one all-reduce of [grad_transform | grad_bias] per
 * step (+ two of the batch-norm statistics when sync_batch_norm). The 128-byte id is ncclUniqueId.
 * SEMANTICS — read before training with world_size > 1: only the dense projection, its bias and the batch-norm
 * statistics are reduced; the word and document tables (and their optimiser state) are updated from each rank's own
 * shard ("sparse embedding rows stay GPU-local"), so the replicas of the tables drift apart and an N-rank run does
 * NOT follow the single-GPU trajectory (the loss and the dense gradients of a step given equal tables do, exactly).
 * nvsm_dp_average_tables replaces every replica's tables by their mean over the ranks; cuNVSMTrainModel calls it at the
 * end of every epoch and before every model dump, so that what rank 0 writes carries every rank's updates. A caller
 * that checkpoints one rank without it drops the other ranks' embedding updates.
 * EXACT TABLES (nvsm_config.dp_exact_tables = 1): the window ids, document ids, projected phrases, multipliers and phrase
 * gradients of every rank are all-gathered (ncclAllGather on the main stream, 2 x B x (d_e + d_w) floats per rank and step)
 * and every rank runs the table updates of the whole global batch in rank order — exactly the single-GPU update on the
 * concatenated batch, so the tables of all ranks are bit-identical after every step and equal the single-GPU tables up to the
 * summation order of the all-reduced dense statistics. Every rank must pass the same num_instances and agree on whether
 * feature_weights is NULL. The step's collectives all run on the main stream in this mode, the table passes are not
 * overlapped with the backward products, and nvsm_dp_average_tables has nothing to do (it returns at once). The update work
 * per rank is that of the global batch: the mode buys the single-GPU trajectory, not update throughput.
 * nvsm_get_cost with world_size > 1 is a collective when called before nvsm_compute_gradients (it all-reduces a copy of
 * the loss word); every rank must make the same sequence of calls.
 * NEGATIVES with world_size > 1: rank r is taken to hold instances [r·B, (r+1)·B) of a global batch of world_size·B.
 * NVSM_SAMPLER_HOST_MINSTD: every rank replays the draws of the WHOLE global batch from its copy of the shared generator
 * (nvsm_rng_set_state: the same state on every rank, as cuNVSMTrainModel hands it over) and keeps its own slice's — the
 * negatives of an instance are those of the single-GPU run, ranks never share a negative set, and the generator states
 * stay equal across ranks (world_size x the host draws per rank: this is the parity sampler). NVSM_SAMPLER_DEVICE: the
 * counter-based sampler is keyed by (seed, rank, step, slot), so ranks draw independent streams. */
int nvsm_comm_unique_id(char id[128]);
int nvsm_comm_init(nvsm_model* m, const char id[128]);
/* ncclCommCount of the handle's communicator (0 = none was built) */
int nvsm_comm_size(nvsm_model* m, int* ranks);
/* collective: W, E ← mean over ranks (synchronises the handle) */
int nvsm_dp_average_tables(nvsm_model* m);
/* Alternative transport for tests: the library hands a HOST double buffer to the callback, which must
 * sum it in place across ranks (e.g. torch.distributed gloo). */
typedef int (*nvsm_allreduce_fn)(double* host_buf, int64_t count, void* user);
int nvsm_set_allreduce_callback(nvsm_model* m, nvsm_allreduce_fn fn, void* user);
/* Single-process check of the RCCL plumbing (dlopen, symbols, enum values, stream use): builds a 1-rank communicator
 * on `device` and all-reduces an f32 and an f64 buffer through the same code path nvsm_step uses with world_size > 1. */
int nvsm_comm_selftest(int device);
/* The three collectives of a data-parallel step (all-reduce of [Σx | Σx²]: 2·entity_dim doubles; of [loss | Σdy | Σdy·x̂]:
 * 1 + 2·entity_dim doubles; of the projection gradient: entity_dim·word_dim floats) on a 1-rank communicator, each on a stream of
 * its own, `repeats` times back to back: average microseconds per call in us[0..2] and the payload bytes in bytes[0..2] — the
 * latency floor of each collective on this GPU (what a rank of the N-GPU job pays per step before any wire time), for
 * bench.py's `--gpus 1` line. */
int nvsm_comm_latency(int device, int entity_dim, int word_dim, int repeats, float us[3], int64_t bytes[3]);

/* Per-kernel timing of the hot path, measured with HIP events on the handle's stream (bench.py's
 * roofline leg). enable=1 records around every launch of subsequent steps (adds sync points at
 * read-out only). nvsm_profile_get returns accumulated milliseconds and launch counts per kernel name;
 * names are listed by nvsm_profile_names (NUL-separated, double-NUL terminated). The ~50 event records of a fully
 * profiled step cost ≈5 % of its time (measured); nvsm_profile_select(names) restricts recording to the comma-separated kernel groups
 * (NULL or "" = all) so that a timed region can carry the roofline kernel's events only. */
int nvsm_profile_enable(nvsm_model* m, int enable);
int nvsm_profile_select(nvsm_model* m, const char* kernel);
int nvsm_profile_reset(nvsm_model* m);
int nvsm_profile_names(nvsm_model* m, char* buf, int64_t buf_bytes);
int nvsm_profile_get(nvsm_model* m, const char* kernel, double* total_ms, int64_t* launches);

/* roctx ranges (the reference's nvtxRangePush / nvtxRangePop, cpp/main.cu:386-431): forwarded to
 * librocprofiler-sdk-roctx when it can be loaded, no-ops otherwise. The library itself brackets ComputeCost /
 * ComputeGradients / UpdateParameters and every kernel group; the trainer adds Epoch / Batch / FetchData. */
void nvsm_range_push(const char* name);
void nvsm_range_pop(void);

/* (The unit-test hooks for single kernels — nvsm_debug_* — are NOT in this library: include/cunvsm_amd_test_hooks.h,
 * libcunvsm_amd_testhooks.so.) */

#ifdef __cplusplus
}
#endif
#endif /* CUNVSM_AMD_H */
