/*
 * libtoricenv -- C-ABI of the MI355X-native batched toric-code environment.
 *
 * This is the drop-in boundary for the reference's env hot path.  The reference has no
 * FFI of its own (pure Python duck typing); each entry point below names the Python
 * interface it replaces (paths relative to the upstream tree).  A reference-side ctypes
 * binding is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (TQ_E_*); tq_last_error() gives
 *    the message of the calling thread's last failure.  Nothing aborts.
 *  - all array arguments are DEVICE pointers owned by the caller (e.g. a PyTorch-ROCm
 *    tensor's data_ptr()) unless the name says "host".  The library owns only the
 *    per-handle lattice state.  No allocation and no synchronisation happens on the hot
 *    path: kernels are enqueued on `stream` (a hipStream_t, NULL = default stream) and
 *    the call returns.
 *  - lattices are (2,d,d): [0] = vertex matrix, [1] = plaquette matrix (src/util.py:63-64);
 *    qubit codes I=0 X=1 Y=2 Z=3 (docs/toric_model.md:11); action = [layer,row,col,op],
 *    op in 1..3 (src/util.py:10, src/numba/util_actor.py:100-104).
 *  - a handle is not thread-safe; use one handle per process / GPU
 *    (one actor process per device: Distributed_mp.py:201-211).  Entry points make the handle's
 *    device current while they run and restore the caller's device before returning.
 *  - set-up calls (tq_create, tq_destroy, tq_set_perror_schedule, tq_states_reserve, the first use of a
 *    lattice size on a device) allocate and synchronise; everything else only enqueues kernels.
 *    tq_create, tq_destroy and tq_states_reserve leave the caller's current device unchanged.
 *  - tq_states_persp_count / tq_states_persp_write use one per-device scratch area sized by
 *    tq_states_reserve (they return TQ_E_CAPACITY when it is too small, they never allocate); calls
 *    that use it must be issued on one stream or be separated by a synchronisation.
 *  - alignment: `actions`, `actions_out`, `offsets`, `counts`, `positions` and the stack `out` are
 *    accessed with 16-byte vector loads/stores and must be 16-byte aligned (TQ_E_INVALID otherwise;
 *    any hipMalloc / PyTorch allocation is).  For full store bandwidth `out` and `positions` should
 *    be 128-byte aligned (tq_persp_write writes whole 128-byte lines).  A row of a 2-D int64 offsets
 *    array needs an even row length.
 *  - RNG: counter-based Philox4x32-10 keyed (seed, global env id, episode, round/step);
 *    contract in DESIGN.md.  Results are identical for any partition of env ids over GPUs.
 */
#ifndef TORICENV_H
#define TORICENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TQ_VERSION 200      /* 200: packed block carries f32 priority; every slot written (op 0 = no transition) */

#define TQ_OK 0
#define TQ_E_INVALID (-1)   /* bad argument (NULL handle, even d, unsupported d, n <= 0 ...) */
#define TQ_E_HIP (-2)       /* a HIP runtime call failed; message in tq_last_error() */
#define TQ_E_CAPACITY (-3)  /* output capacity too small */
#define TQ_E_ACTION (-4)    /* an action outside the lattice / op not in 1..3 was seen on the device */
#define TQ_E_INDEX (-5)     /* tq_reset_idx saw an index out of range or listed twice (latched, tq_check) */
#define TQ_E_RESET (-6)     /* a reset hit the round limit without producing a defect (latched, tq_check) */

/* element type of the perspective stack written by tq_persp_write */
#define TQ_F32 0            /* float32 -- what the reference feeds the NN (numba/util_actor.py:39) */
#define TQ_F16 1
#define TQ_BF16 2
#define TQ_U8 3

/* p_error strategy of the fused auto-reset (Actor_mp.py:176-180) */
#define TQ_PERR_FIXED 0     /* every reset uses p_error_default */
#define TQ_PERR_LINEAR 1    /* roof = min(final, roof + delta); p = roof */
#define TQ_PERR_RANDOM 2    /* roof as above; p ~ U(start, roof) */

typedef struct tq_env tq_env;

int tq_version(void);
const char* tq_last_error(void);

/* gym.make('toric-code-v0', config={"size","min_qubit_errors":0,"p_error"}) + EnvSet(env, no_envs)
 * (Distributed_mp.py:72-76, src/EnvSet.py:5-16).  d odd in {3,5,...,21}.  Lattice e of this
 * handle has global env id first_env_id + e (shard offset for multi-GPU). */
int tq_create(tq_env** out, int n_envs, int d, int device, uint64_t seed, int64_t first_env_id);
int tq_destroy(tq_env* h);

/* env config: default p_error in (0,1] (gym config "p_error"), terminal reward (default 100,
 * evaluation.py:175), max_actions_per_episode (default 75, Distributed_mp.py:44; used only by
 * tq_actor_step's auto-reset).  p_error = 0 is rejected (a reset could never produce a defect) unless the
 * fixed-n sampler was selected first with tq_set_min_qubit_errors(n > 0), which does not use p_error. */
int tq_set_params(tq_env* h, double p_error_default, double terminal_reward, int max_steps_per_episode);
/* gym config "min_qubit_errors": 0 (default, every config of the reference tree) = depolarizing
 * sampler at p_error; n > 0 = every reset places exactly n errors on uniformly chosen distinct
 * qubits with uniform Paulis (the fixed-n sampler, results/small_p_error_test.py:34-40; p_error
 * is then unused), still redrawn until the syndrome is non-empty. */
int tq_set_min_qubit_errors(tq_env* h, int n_errors);
/* p_error schedule of the actor's reset policy (Actor_mp.py:41-46,176-180) for tq_actor_step. */
int tq_set_perror_schedule(tq_env* h, int strategy, double p_start, double p_final, double p_delta);

/* Set-up helper for the caller-owned stack buffer (np.concatenate's result, numba/util_actor.py:37-39, written every
 * step).  On MI355X a buffer has a write rate of its own for every write stream into it -- the stack write, a plain
 * fill, hipMemset -- between 5.2 and 6.9 TB/s, different from allocation to allocation and from box to box; plain
 * hipMalloc / torch.empty buffers were the slow kind in every PyTorch process of round 3 (5.1-5.5 TB/s), buffers made
 * of 2 MiB physical chunks (HIP virtual memory API) the fast kind in most (profiles/r03_stack_write_ab.txt).
 * tq_stack_alloc returns device memory of at least `bytes` bytes made of such chunks, each mapped once behind one
 * virtual range, zero-filled, and CHECKED: every 2 MiB page is verified to be reached through its own address (this
 * API leaves stale address translations behind when addresses are re-used; the library never re-uses one, never gives
 * an address range back, and does not hand out a buffer that fails the check).  Allocates, maps and synchronises; any
 * other device allocation works as `out` of tq_persp_write just as well. */
int tq_stack_alloc(int device, uint64_t bytes, void** out);
/* Gives the physical chunks back (synchronises the device first).  The VIRTUAL ADDRESS RANGE IS NEVER RETURNED: by
 * design every tq_stack_alloc leaks its 2 MiB-rounded address range for the life of the process (a re-used range
 * reaches the previous tenant's pages through stale translations on ROCm 7.2), out of a 128 TiB address space --
 * a set-up call, not something to call per step. */
int tq_stack_free(void* ptr);

/* The stack write runs one persistent workgroup per CU, each with a fixed contiguous share of the stack, dealt to the
 * XCDs round-robin.  On MI355X the CUs of the odd XCDs store this stream ~20 % slower than those of the even ones (every
 * box and buffer measured: profiles/r04_workgroup_end_times.txt), so of every pair of workgroups the even one takes
 * 32 + bias and the odd one 32 - bias of the pair's 64 fine parts -- for d >= 7, f32 / f16 / bf16 stacks of 64 MB and more
 * (smaller lattices and the u8 stack are bound by the producers, not by the stores: there unequal shares only cost).  Default 5 (or the
 * environment variable TORICENV_XCD_BIAS, read once); 0 = equal shares; changes the speed of tq_persp_write*, never
 * its result.  tq_set_xcd_bias is the process-wide setting, tq_env_set_xcd_bias one handle's own (-1 = follow the
 * process-wide one, the default).  EnvSet.pickStackBuffer times the caller's own write both ways and sets the HANDLE
 * to the faster. */
int tq_set_xcd_bias(int bias);   /* 0..16, else TQ_E_INVALID */
int tq_get_xcd_bias(void);
int tq_env_set_xcd_bias(tq_env* h, int bias);   /* -1..16 */
int tq_env_get_xcd_bias(const tq_env* h);       /* the setting in force for this handle */

int tq_num_envs(const tq_env* h);
int tq_size(const tq_env* h);

/* EnvSet.resetAll(p_errors) (EnvSet.py:29-36): p_err = device f64[N] or NULL (default p). */
int tq_reset_all(tq_env* h, const double* p_err, void* stream);
/* EnvSet.resetTerminalEnvs(idx, p_errors) (EnvSet.py:19-27): idx = device i32[n_idx] (distinct,
 * in range), p_err = device f64[n_idx] or NULL.  Checked on the device: an index out of range is
 * skipped, of an index listed twice only one copy resets the lattice, and TQ_E_INDEX is latched. */
int tq_reset_idx(tq_env* h, const int32_t* idx, int n_idx, const double* p_err, void* stream);

/* EnvSet.step(actions) (EnvSet.py:38-47): actions = device i32[N,4]; rewards f32[N];
 * terminals u8[N].  No auto-reset (the caller resets, Actor_mp.py:171-183).  The pre-step
 * syndrome is kept inside the handle for tq_transition_write. */
int tq_step(tq_env* h, const int32_t* actions, float* rewards, uint8_t* terminals, void* stream);

/* env.state / EnvSet.states: syndrome as u8[N,2,d,d] (0/1). */
int tq_get_state(tq_env* h, uint8_t* out, void* stream);
/* rows idx[0..n_idx) only -> u8[n_idx,2,d,d] (return value of resetTerminalEnvs). */
int tq_get_state_idx(tq_env* h, const int32_t* idx, int n_idx, uint8_t* out, void* stream);
/* env.qubit_matrix as u8[N,2,d,d] Pauli codes. */
int tq_get_qubits(tq_env* h, uint8_t* out, void* stream);
/* overwrite env.qubit_matrix and recompute env.state = createSyndromOpt(qubit_matrix)
 * (results/small_p_error_test.py:112-120, results/start_from_state.py:34-38). */
int tq_set_qubits(tq_env* h, const uint8_t* qubits, void* stream);
/* per-lattice counters: episodes started, steps taken in the current episode (u32[N] each). */
int tq_get_counters(tq_env* h, uint32_t* episodes, uint32_t* steps, void* stream);
/* env.evalGroundState() per lattice -> u8[N] (1 = no non-trivial loop). */
int tq_eval_ground_state(tq_env* h, uint8_t* out, void* stream);
/* env.isTerminalState(state) per lattice -> u8[N]. */
int tq_is_terminal(tq_env* h, uint8_t* out, void* stream);

/* generatePerspectiveBatch, step 1 (numba/util_actor.py:56-67 + cumsum :35): counts i32[N]
 * (may be NULL) and offsets i64[N+1] (exclusive scan, offsets[N] = P).  A by-product are the cut points of the batch
 * into 8192 fine parts of equal perspective count from which tq_persp_write(the same `offsets` pointer) makes its 256
 * workgroups' shares; the handle keeps the tables of the last TWO calls (a write that is still running on another stream
 * reads the older one). */
int tq_persp_count(tq_env* h, int32_t* counts, int64_t* offsets, void* stream);
/* generatePerspectiveBatch + np.concatenate, step 2 (numba/util_actor.py:33-39): writes the
 * env-major stack out[P,2,d,d] of element type `dtype` and positions i32[P,3] (may be NULL)
 * for the offsets from tq_persp_count.  capacity = number of perspectives `out` can hold;
 * lattices that would overflow it are skipped and TQ_E_CAPACITY is latched (tq_check).
 * `offsets` must be the scan of the lattices' CURRENT perspective counts (tq_persp_count after the last call that
 * changed a syndrome).  The device checks it: offsets that are not (stale, shifted, all zero, decreasing, from another
 * batch) are refused -- the kernel stores nothing outside [0, min(offsets[N], capacity)) perspectives, every wait in
 * it is bounded, the grid drains, and TQ_E_INVALID is latched for tq_check; the handle stays usable.
 * At most EIGHT stack writes of one handle may be in flight at a time (on whatever streams): their workgroups take their
 * shares through a ring of eight counter sets (tq_set_xcd_bias); a write captured into a HIP graph may be replayed. */
int tq_persp_write(tq_env* h, const int64_t* offsets, void* out, int32_t* positions,
                   int64_t capacity, int dtype, void* stream);
/* The same for the lattices [first, first + count) only: `out` / `positions` receive the perspectives
 * of those lattices, the first one at index 0 (`offsets` is still the whole batch's scan), so a
 * consumer with a small buffer -- the NN forward of numba/util_actor.py:39-46 works in chunks anyway --
 * can walk a batch whose whole stack it does not want to hold (d=9: 5.3 GB per 65 536 lattices). */
int tq_persp_write_range(tq_env* h, const int64_t* offsets, int first, int count, void* out,
                         int32_t* positions, int64_t capacity, int dtype, void* stream);

/* Same two steps for a batch of syndromes that does not live in a handle (the learner's
 * predictMaxOptimized, util_learner.py:48-111): states = device u8[n,2,d,d]. */
/* set-up: size the calling device's scratch for up to n_max states of size d (allocates, synchronises).  The
 * tq_states_persp_* calls share that ONE scratch per device: use them from one stream per device at a time. */
int tq_states_reserve(int d, int n_max);
int tq_states_persp_count(int d, int n, const uint8_t* states, int32_t* counts, int64_t* offsets,
                          void* stream);
int tq_states_persp_write(int d, int n, const uint8_t* states, const int64_t* offsets, void* out,
                          int32_t* positions, int64_t capacity, int dtype, void* stream);

/* _selectActionBatch_prime (numba/util_actor.py:69-107) on the device: q_table f32[P,3],
 * offsets i64[N+1], positions i32[P,3], eps f64[N] -> actions i32[N,4], q_values f32[N,3].
 * greedy iff (1-eps) > U; greedy = first maximum in row-major order; otherwise a uniform
 * perspective and op (Philox, keyed by the lattice's episode and step counters).
 * q_table may be NULL when every eps is 1 (pure exploration: the Q-values are never read). */
int tq_select_action(tq_env* h, const float* q_table, const int64_t* offsets,
                     const int32_t* positions, const double* eps, int32_t* actions,
                     float* q_values, void* stream);

/* The same selection for an explicit batch of states that does not live in a handle --
 * selectActionBatch(number_of_actions, epsilon, grid_shift, toric_size, state, model, device)
 * (numba/util_actor.py:11-53) after the model forward.  The reference draws from numpy's global,
 * never seeded RNG (:49, :97-98); here the draw of state i is Philox keyed (seed, first_id + i,
 * call_counter) in its own domain, so the caller supplies a seed and a counter it advances per call. */
int tq_states_select_action(int n, const float* q_table, const int64_t* offsets,
                            const int32_t* positions, const double* eps, uint64_t seed,
                            uint64_t call_counter, int64_t first_id, int32_t* actions,
                            float* q_values, void* stream);

/* Reads and clears the calling device's error latch of the tq_states_* entry points (synchronises
 * `stream`): 0, TQ_E_ACTION, TQ_E_CAPACITY or TQ_E_INVALID (offsets that do not belong to the states). */
int tq_states_check(void* stream);

/* predictMaxOptimized's reduction (util_learner.py:96-110): out[i] = max over the (n_i,3) slice of
 * q_table, 0 for states without perspectives; `largest` = device i32[1] holding the longest slice
 * length reproduces the reference's zero padding (shorter slices get max(max_q, 0)); NULL = plain max. */
int tq_segment_max(const float* q_table, const int64_t* offsets, int n, const int32_t* largest,
                   float* out, void* stream);

/* generateTransitionParallel (util_actor.py:223-264) for the last tq_step: perspective of the
 * pre-step and post-step syndrome centred on the acted qubit (rotated for layer 1), action
 * rewritten to (layer, gs, gs, op).  Outputs (any may be NULL): persp u8[N,2,d,d],
 * next_persp u8[N,2,d,d], actions_out i32[N,4]. */
int tq_transition_write(tq_env* h, const int32_t* actions, uint8_t* persp, uint8_t* next_persp,
                        int32_t* actions_out, void* stream);

/* generateTransitionParallel(action, reward, state, next_state, terminal, grid_shift, type)
 * (util_actor.py:223-264) for explicit syndrome arrays that do not live in a handle:
 * states / next_states = device u8[n,2,d,d], actions = device i32[n,4]
 * -> persp u8[n,2,d,d], next_persp u8[n,2,d,d], actions_out i32[n,4] (any may be NULL). */
int tq_states_transition(int d, int n, const uint8_t* states, const uint8_t* next_states,
                         const int32_t* actions, uint8_t* persp, uint8_t* next_persp,
                         int32_t* actions_out, void* stream);

/* Packed transition block (the wire format gathered to the replay memory -- the (transition,
 * priority) pairs of Actor_mp.py:152, IO_mp.py:60-66): for `cap` transitions, SoA sections in this
 * order, each 8-byte aligned:
 *   persp_v u64[W][cap] | persp_p u64[W][cap] | next_v u64[W][cap] | next_p u64[W][cap] |
 *   action u32[cap] (layer | row<<8 | col<<16 | op<<24) | reward f32[cap] | priority f32[cap] |
 *   terminal u8[cap]
 * with W = ceil(d*d/64) and bit r*d+c of a plane = cell (r,c); 45 bytes per transition at d=7.
 * A slot whose action word is 0 (op = 0) holds no transition (the lattice was given a no-op or a
 * rejected action): all its other fields are zero and consumers drop it. */
int64_t tq_transition_block_bytes(int d, int64_t cap);
/* unpack slots [first, first+count) of a block into u8 grids / i32 actions (op 0 = empty slot) /
 * f32 rewards / u8 terminals / f32 priorities (any output may be NULL). */
int tq_transition_unpack(int d, const void* block, int64_t cap, int64_t first, int64_t count,
                         uint8_t* persp, uint8_t* next_persp, int32_t* actions, float* rewards,
                         uint8_t* terminals, float* priorities, void* stream);
/* computePrioritiesParallel (util_actor.py:268-287) into the block's priority section, for a block
 * that holds n_steps steps of n_envs lattices in slot order t*n_envs + e (what tq_actor_step writes):
 *   priority = | reward + discount * max_a Q[t+1][e][a] - Q[t][e][op-1] |   (f64 arithmetic, stored f32)
 * q_values = device f32[n_steps+1][n_envs][3]: the q_values selectActionBatch returned at each of the
 * n_steps steps plus the step after (local_buffer_Q and its np.roll(-1), Actor_mp.py:146-150);
 * NULL = all zeros (pure exploration).  Empty slots get priority 0.
 * Width: the reference's priorities are float64 (util_actor.py:287); the wire field is float32 -- the one
 * field of the record that is narrower than upstream.  The arithmetic is f64 and rounded once, so the
 * stored value is exactly float32(reference priority); a sum tree that needs f64 widens it on ingest. */
int tq_block_priorities(int d, void* block, int64_t cap, int n_envs, int n_steps,
                        const float* q_values, double discount, void* stream);

/* One fused iteration of the actor loop body after the policy (Actor_mp.py:116-183):
 *   step (EnvSet.step) -> transition record (generateTransitionParallel) -> reset of terminal
 *   / timed-out lattices with the p_error schedule (resetTerminalEnvs) -> perspective counts
 *   of the resulting states.
 * actions: device i32[N,4] from tq_select_action / the caller, or NULL = pure exploration
 * (eps = 1: uniform random perspective and op drawn in-kernel with the same Philox draws as
 * tq_select_action).  Outputs (may be NULL): actions_out i32[N,4] (the actions taken),
 * rewards f32[N], terminals u8[N], block/slot: packed transition block with capacity
 * `block_cap` transitions, lattice e writing slot `slot_base + e` (always: an empty slot when it
 * was given a no-op or a rejected action; the priority section is left to tq_block_priorities).
 *
 * Two streams: the step reads the lattices from one buffer of the handle and writes the other (the two take turns),
 * and tq_persp_count keeps two cut-point tables in turn.  So with actions == NULL (the selection does not look at the
 * stack) the caller may enqueue tq_actor_step(t) and tq_persp_count(t+1, into a SECOND offsets array) on another stream
 * than tq_persp_write(t): they run beside the stack write instead of behind it.  The caller orders what the handle
 * cannot see: tq_persp_write(t) behind tq_persp_count(t); tq_actor_step(t+1) behind tq_persp_write(t) (it overwrites
 * the buffer that write reads).  On one stream nothing changes.  The set-up and in-place entry points (reset, step,
 * set_qubits, getters) work on the current buffer in stream order like before. */
int tq_actor_step(tq_env* h, const int32_t* actions, int32_t* actions_out, float* rewards,
                  uint8_t* terminals, void* block, int64_t block_cap, int64_t slot_base,
                  void* stream);

/* Reads and clears the handle's device error latch (synchronises `stream`): 0, TQ_E_ACTION,
 * TQ_E_CAPACITY, TQ_E_INDEX, TQ_E_RESET or TQ_E_INVALID (offsets handed to tq_persp_write that are not the scan of the
 * lattices' counts). */
int tq_check(tq_env* h, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TORICENV_H */
