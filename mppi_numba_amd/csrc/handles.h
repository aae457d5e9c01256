// handles.h -- what the opaque handles of include/mppi_hip.h hold (included by mppi_api.hip only).
#pragma once
#include "host_common.h"

// ---------------------------------------------------------------------------
// TDM
// ---------------------------------------------------------------------------
struct mppi_tdm {
  mppi_tdm_cfg cfg;
  hipStream_t stream = nullptr;
  int8_t* grid = nullptr;  // [G][max_rows][max_cols] int8 (reference layout)
  int8_t* pmf = nullptr;   // [B][rows][cols]
  size_t pmf_capacity = 0;
  int8_t* table = nullptr;  // [B] bin -> int8 traction
  int table_capacity = 0;
  int8_t* obs = nullptr;  // [rows][cols]
  int8_t* unk = nullptr;
  int8_t* risk = nullptr;
  size_t map_capacity = 0;
  // staging of the raw inputs of mppi_tdm_set_maps_from_pmf (device-side preprocessing)
  int8_t* raw = nullptr;  // raw PMF | raw obstacle | raw unknown
  size_t raw_capacity = 0;
  float* bin_values = nullptr;
  int bin_values_capacity = 0;
  int* prep_flags = nullptr;
  uint64_t* states = nullptr;  // xoroshiro-compatible generator only
  long n_states = 0;
  int bins = 0, rows = 0, cols = 0;
  bool has_risk = false, maps_set = false, one_hot = false;
  bool compact_ok = false;  // masks are 0/1 and every traction byte is in [0,127]: 16-bit cells usable
  int table_max = 127;      // largest traction byte the sampler can write
  double lo = 0.0, ratio = 0.0;
  uint64_t epoch = 0;         // Philox call counter
  uint64_t maps_version = 0;  // bumped by set_maps
  uint64_t grid_version = 0;  // bumped whenever `grid` changes
  uint64_t sampled_maps_version = ~0ULL;
  double sampled_alpha = -1.0;
  // solve() of a CVaR planner samples straight into the planner's cell words (Philox only):
  // the int8 grids are then produced on demand from the same counters
  bool grid_stale = false;      // `grid` does not hold the draws of (sampled_epoch, sampled_alpha) yet
  uint64_t sampled_epoch = 0;   // Philox epoch of the current draws
  bool injected = false;  // grids came from mppi_tdm_set_sampled_grids
  int8_t injected_max = 0, injected_min = 0;
  // How many concentric rings of border cells carry ZERO traction in every grid a sampler can write (the padding ring of
  // terrain.py:511-583: a rollout that enters it never moves again, so none leaves the map as long as a step is no longer
  // than the ring is wide).  The exact schedule's state role computes its LDS address without a clamp only on maps where
  // that -- or the reach bound -- proves the clamp idle (launch_plan.h: unclamped_lookup_ok); capped at kSinkRingCap.
  static constexpr int kSinkRingCap = 16;
  int maps_sink_ring = 0, injected_sink_ring = 0;
  // samples sharded over GPUs (mppi_tdm_set_sample_shard): this handle's G grids are samples
  // [first_sample, first_sample + G) of the unsharded set; even, a Philox block serves a pair
  int first_sample = 0;
};

// ---- k_rollout_scan / k_rollout_scan_exact: the launch geometry (launch_plan.h: scan_plan) --------
struct ScanPlan {
  int waves = 0;       // waves per workgroup: one per 8 steps (+ the three walkers of the exact kernel)
  int chunk_waves = 0; // ... of which work on 8 steps each
  int tile = 32;       // rollouts per workgroup: 32 (two lanes per rollout) or 64
  size_t lds = 0;
  bool pow2res = false;
  bool exact = false;  // k_rollout_scan_exact: the three running sums walked with the reference's roundings
  // k_rollout_scan_exact on a map the planner has stopped speculating on (round 5): every tile runs the exact
  // three-wave schedule at once (ScanFallback::direct) -- noise, folded update and tile packets as ever, so the
  // iteration stays one launch and the kernel family does not depend on the map
  bool direct = false;
  int fallback_offset = -1, fallback_map_bytes = 0, small_offset = 0;  // where the exact schedule's controls | window | ring and the small arrays live (direct)
};

// ---------------------------------------------------------------------------
// planner
// ---------------------------------------------------------------------------
struct mppi_planner {
  mppi_planner_cfg cfg;
  hipStream_t stream = nullptr;
  int n_local = 0, n_offset = 0;
  // batched multi-query: B problems, each n_inst rollouts (inst_tiles tiles of 64) on this GPU;
  // n_local = B * n_inst.  Per-problem start / goal / window origin live in inst_dev.
  int B = 1, n_inst = 0, inst_tiles = 0;
  std::vector<BatchInst> inst_host;
  BatchInst* inst_dev = nullptr;
  bool inst_set = false, inst_dirty = false;
  // pinned, device-mapped (B,T) mirror of u: the update kernels write it (u_host_dev is the device
  // view of the same memory), solve() reads it after the stream has drained -- no copy on the hot path
  float2* u_host = nullptr;
  float2* u_host_dev = nullptr;
  // Only solve() reads the mirror, after its LAST iteration: the update launches of every other
  // iteration are spared the posted write across PCIe (their completion waits for it).
  bool mirror_now = false;   // the coming update launch writes the mirror
  bool mirror_done = false;  // ... the last one did
  // set_u(): pinned staging + asynchronous copy; the next set_u waits for the previous copy only
  float2* u_stage = nullptr;
  hipEvent_t ev_u_staged = nullptr;
  bool u_stage_busy = false;
  // device buffers
  float2* noise = nullptr;    // tile-major (n_local, T): the buffer the NEXT rollout/update reads
  float2* noise_buf[2] = {nullptr, nullptr};  // double buffer: noise of iteration k+1 is generated
  int noise_cur = 0;                           // while iteration k runs (in-launch or on noise_stream)
  // throughput regime (no idle CU for in-launch generation): the noise of iteration k+1 runs beside
  // rollout k on a second stream while the rollout leaves wave slots free (run_iterations)
  hipStream_t noise_stream = nullptr;
  hipEvent_t ev_buf_free = nullptr, ev_noise_ready = nullptr;
  // ... and how the consumer of that noise is ordered behind it (DevParams::noise_flag): a sequence number stored by a
  // one-thread kernel behind every generator on noise_stream; a rollout kernel that looks at it itself takes the place of
  // the cross-stream wait, every other launch gets the wait (launch_plan.h: settle_noise_wait)
  unsigned long long* noise_flag_dev = nullptr;
  unsigned long long noise_flag_seq = 0;     // generators launched on noise_stream so far
  unsigned long long noise_flag_expect = 0;  // ... the one whose noise the next rollout launch reads
  bool noise_wait_pending = false;           // that launch has not been ordered behind its noise yet
  // ... and how the generator is held back until the buffer it overwrites is free (DevParams::progress): launches that
  // signal their start let a one-wave gate kernel on noise_stream take the place of ev_buf_free
  unsigned long long* progress_dev = nullptr;
  unsigned long long progress_seq = 0;       // launches that signal, so far
  bool progress_signalled = false;           // the rollout launch of this iteration does
  bool progress_capable_last = false;        // ... the previous one did: no event is recorded in front of the next
  bool update_signals = false;               // the coming k_update_rows launch is to signal its start (generator beside the update)
  bool update_signalled = false;             // ... and it did (that kernel ran)
  unsigned int* flag_fault_host = nullptr;   // pinned, device-mapped: a bounded flag wait gave up (DevParams::flag_fault)
  unsigned int* flag_fault_dev = nullptr;
  bool stream_flags_off = false;             // ... after which this handle orders its two streams with events
  // hipGraph replay of the iteration loop (mppi_planner_set_graph_replay): two iterations
  // (one round of the noise double buffer) captured once, replayed while nothing a kernel
  // argument carries has changed.  See run_iterations.
  bool graph_on = false;
  int graph_chunk = 2;                    // iterations per captured graph
  unsigned long long* gen_dev = nullptr;  // device: update kernels executed since graph mode was enabled
  uint64_t bumps_launched = 0;            // host mirror of *gen_dev once the stream has drained
  bool primed = false;                    // noise_buf[noise_cur ^ 1] already holds the NEXT iteration's noise
  bool graph_warm = false;                // one direct iteration has run since graph mode was enabled
  // one cached graph per parity of the noise double buffer and of the two control buffers of a sharded
  // handle (a call with an odd number of iterations leaves the other parity behind)
  // (... and of the two tile packet buffers of the time-parallel kernels)
  static constexpr int kGraphSlots = 8;
  hipGraph_t graph[kGraphSlots] = {};
  hipGraphExec_t graph_exec[kGraphSlots] = {};
  std::vector<unsigned char> graph_sig[kGraphSlots];  // everything the captured launches took by value
  uint64_t graph_spec_launches[kGraphSlots] = {};  // speculative launches in one replay of the graph
  int graph_u_flip[kGraphSlots] = {}, graph_tpk_flip[kGraphSlots] = {};  // buffer parities one replay changes
  int u_parity = 0;                         // flips whenever u and u_alt change places
  long graph_replays = 0, graph_captures = 0;
  std::string last_rollout;        // which rollout kernel variant the last launch used (diagnostic)
  int debug_flags = 0;             // mppi_planner_set_debug_flags (tests pin every kernel variant through it)
  bool next_noise_wanted = false;  // the coming rollout launch should also generate noise_buf[cur^1]
  bool next_noise_done = false;    // ... and it did
  bool noise_on_side_stream = false;  // the noise produced ahead is still in flight on noise_stream
  float2* staging = nullptr;  // (n_local,T) host-layout staging for set/get_noise
  float2* u = nullptr;        // [T]
  float2* u_prev = nullptr;   // [T]
  float2* u_alt = nullptr;    // [T] the other control buffer of a sharded handle (launch_apply, PendingApply)
  bool apply_pending = false; // the all-gathered packets hold an update the next rollout launch applies
  uint64_t folded_applies = 0;
  float* costs = nullptr;     // [n_local]
  float* weights_out = nullptr;  // [n_local] normalised weights, filled on request
  float* w_rel = nullptr;      // [n_local] exp(-(c - beta_tile)/lambda)
  float* tile_beta = nullptr;  // [n_tiles] minimum cost of each tile of 64 rollouts
  int n_tiles = 0;
  bool tile_packets_fresh = false;  // w_rel / tile_beta written by the rollout kernel for the current costs
  bool theta_bounded = false;  // |heading| < 5e4 rad over the horizon (decided per launch: launch_rollout)
  // The time-parallel kernels (rollout_scan*.h) leave one packet per tile of 32 (64) rollouts -- minimum cost,
  // sum of w_rel, sum of w_rel * noise(t) -- consumed by k_combine_tiles or, inside an iteration loop on one GPU,
  // by the class reduction of the NEXT rollout launch (update_kernels.h, PendingApply::reduce_tiles): two buffers,
  // a launch reads its predecessor's while it writes its own.  Allocated with the handle (a lazy hipMalloc could
  // fall inside a stream capture).
  float* tile_packets[2] = {nullptr, nullptr};  // [ceil(n_local / 32)][tile_packet_floats(T)]
  int tpk_cur = 0;         // the buffer the last such launch wrote
  int scan_tile = 32;      // rollouts per tile of the last such launch
  bool scan_packets_fresh = false;  // ... written by the last rollout launch for the current costs
  unsigned long long* published = nullptr;  // [2][T][kPublishedStride] words, one used per step: {float u.x; float u.y}, kNotPublished (all ones) until written and between loops (planner_alloc, k_combine_tiles): update_kernels.h
  // the in-launch hand-over of the controls is bounded and fails soft (update_kernels.h, PendingApply::fault)
  unsigned int* fold_fault_host = nullptr;  // pinned, device-mapped: a workgroup gave up waiting for a published step
  unsigned int* fold_fault_dev = nullptr;
  int fold_max_polls = 1 << 20;             // ... after this many polls (~a second)
  long iterations_since_wait = 0;           // enqueued since the host last waited for the stream (wait_for_stream)
  bool fold_off = false;                    // ... after which this handle updates through launches of their own
  uint64_t fold_faults = 0;
  bool reduce_pending = false;  // the last launch's tile packets hold an update that the next rollout launch applies
  int reduce_index = 0;         // such launches since the loop began: picks the flag set
  uint64_t reduced_applies = 0;
  uint64_t bumps_owed = 0;      // graph replay: iterations whose update kernel (and its epoch bump) did not run
  // the iteration loop of such a handle generates the noise INSIDE the rollout launch (Philox counter
  // blocks, never stored): noise_buf is then stale, and whoever wants the noise of the last iteration
  // (get_noise, get_state_rollout, a stage-level update) has it regenerated from the same counters
  bool scan_gen_now = false;   // the coming rollout launch is to generate its own noise
  bool noise_virtual = false;  // the noise of the last iteration exists as counters only ...
  int noise_virtual_back = 1;  // ... of Philox epoch noise_epoch - noise_virtual_back
  double* packets = nullptr;  // [world][2+2T]; own packet at [rank]
  double* stats = nullptr;    // {beta, den} of the last update
  uint32_t* cells = nullptr;
  size_t cells_capacity = 0;
  double* cc_scratch = nullptr;  // [T][n_local] control-cost products of the pipelined rollout
  uint16_t* cells16 = nullptr;  // 16-bit cells, row pitch multiple of 8 (LDS window source)
  size_t cells16_capacity = 0;
  int pitch16 = 0;
  bool cells16_valid = false;
  bool cells16_with_risk = false;  // cells16 holds 32-bit cells with the risk byte (speed-map mode)
  int num_cus = 256;
  int lds_per_cu = 160 * 1024;
  int8_t* risk_ref = nullptr;
  float* sample_costs = nullptr;  // [n_local][M], allocated on first request
  bool want_sample_costs = false;
  uint64_t* states = nullptr;  // xoroshiro-compatible generator only
  long n_states = 0;
  float2* obs_pos = nullptr;
  float* obs_r = nullptr;
  int n_obstacles = 0;
  std::vector<float> obs_pos_host, obs_r_host;  // what the device arrays hold (mppi_planner_set_disc_obstacles)
  float* state_rollout = nullptr;  // [V][T+1][3]
  // host state
  mppi_params params;
  bool params_set = false;
  uint64_t noise_epoch = 0;
  const mppi_tdm* packed_lin = nullptr;
  const mppi_tdm* packed_ang = nullptr;
  uint64_t packed_lin_grid = ~0ULL, packed_ang_grid = ~0ULL, packed_lin_maps = ~0ULL;
  // timing
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_stage[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool profile_stages = false;
  float stage_ms[4] = {0, 0, 0, 0};
  float last_elapsed_ms = 0.f;
  bool elapsed_pending = false;
  int last_iterations = 0;
  // Speculative rollout kernels (the time-parallel k_rollout_scan*) on a map where the traction
  // changes from cell to cell: every tile fails its vote and re-runs on the exact schedule, slower
  // than launching k_rollout_pipe in the first place (N = 8192, T = 200 over a CVaR-bin map: 85 vs
  // 49 us).  The kernels count failed tiles in a host-mapped word; whenever the host has
  // synchronised anyway it compares that with the LAUNCHES it made (a launch lasts as long as its slowest
  // tile: one failing tile stalls it) and, from one failed tile per two launches on, stops speculating
  // until the packed map changes.
  // scan_plan()'s answer for the state it was derived from (launch_plan.h)
  std::vector<unsigned char> scan_key;
  ScanPlan scan_cached;
  bool scan_cached_ok = false;
  unsigned int* spec_fail_host = nullptr;  // pinned, device-mapped
  unsigned int* spec_fail_dev = nullptr;   // device view of the same word
  uint64_t spec_launches = 0;
  bool speculation_off = false;
  // mppi_planner_time_kernels: dispatch begin / end of the rollout and update launches of the
  // iterations it runs (4 events per iteration), picked up by MPPI_KLAUNCH
  hipEvent_t kev_start = nullptr, kev_stop = nullptr;
  std::vector<hipEvent_t> ktime_events;
  std::vector<char> ktime_update_ran;  // per timed iteration: its update was a launch of its own
  int ktime_index = -1;
  bool ktime_markers = false;  // a timed rollout launch followed a cross-stream wait / an event record on the stream
  unsigned long long* ktime_dev = nullptr;  // [reps][4][ktime_waves] when the waves of the timed rollout | update launches entered, left (DevParams::ktime)
  size_t ktime_dev_capacity = 0;            // ... in words
  int ktime_waves = 0;
  bool used_side_stream = false;   // an iteration put the next one's noise on noise_stream (since the flag was last cleared)
  bool ktime_use_stamps = false;   // this timing run is a plain loop whose kernels stamp themselves
  unsigned long long* ktime_rollout_slot = nullptr;  // this iteration's slots, while its launches are made
  unsigned long long* ktime_update_slot = nullptr;
  // comm
  ncclComm_t comm = nullptr;
  // The peer exchange (mppi_planner_p2p_*; update_kernels.h, PeerExchange): this rank's inbox -- fine-grained device
  // memory the other ranks write into -- and the other ranks' inboxes as this device addresses them (peer access within
  // one process, IPC handles across processes).  With it a sharded iteration of the time-parallel exact kernel is ONE
  // launch and no collective; the last iteration of a call exchanges inside k_combine_tiles.
  unsigned long long* inbox = nullptr;
  unsigned long long* peer_inbox[kMaxFoldedRanks] = {};
  bool peer_mapped[kMaxFoldedRanks] = {};  // opened with hipIpcOpenMemHandle: to be closed
  const char* inbox_kind = "";  // how the inbox was allocated (diagnostic)
  unsigned int* p2p_fault_host = nullptr;  // pinned, device-mapped: a peer's numbers did not arrive (exchange_step)
  unsigned int* p2p_fault_dev = nullptr;
  bool p2p_on = false;
  int p2p_index = 0;  // exchanges so far: its parity picks the inbox set (the same on every rank)
  uint64_t p2p_exchanges = 0;
  // CVaR mode with the M traction samples sharded over GPUs (mppi_planner_set_sample_sharding):
  // this handle rolls ALL N control samples over its cfg.num_grid_samples grids; the per-(n, m)
  // costs of all shards are all-gathered and every rank forms the CVaR of every control sample
  int m_rank = 0, m_count = 1;
  // a stage-level rollout of such a handle leaves the CVaR over the LOCAL samples in costs: the update
  // must not run before the slabs of all shards have been reduced (launch_cvar_reduce)
  bool sample_costs_local_only = false;
  float* slabs = nullptr;  // [m_count][n_local][M_local]
  // closed loop on the device (mppi_planner_closed_loop): world state, trajectory log
  double* loop_state = nullptr;   // [B][3]
  double* loop_xhist = nullptr;   // [B][loop_capacity + 1][3]
  float2* loop_uhist = nullptr;   // [B][loop_capacity]
  int* loop_done = nullptr;       // [B]
  float2* loop_u_final = nullptr; // [B][T] controls of a problem at the step it reached its goal
  int* loop_done_count = nullptr;      // pinned, device-mapped
  int* loop_done_count_dev = nullptr;  // device view of the same int
  int loop_capacity = 0;
};

// rings of cells, from the border inwards, for which sink(row, col) holds (see mppi_tdm::maps_sink_ring)
template <typename F>
static int count_sink_rings(int rows, int cols, F&& sink) {
  int rings = 0;
  for (; rings < mppi_tdm::kSinkRingCap && 2 * rings < rows && 2 * rings < cols; ++rings) {
    const int r0 = rings, r1 = rows - 1 - rings, c0 = rings, c1 = cols - 1 - rings;
    bool all = true;
    for (int c = c0; c <= c1 && all; ++c) all = sink(r0, c) && sink(r1, c);
    for (int r = r0; r <= r1 && all; ++r) all = sink(r, c0) && sink(r, c1);
    if (!all) break;
  }
  return rings;
}

static void drop_graphs(mppi_planner* p) {
  for (int i = 0; i < mppi_planner::kGraphSlots; ++i) {
    if (p->graph_exec[i]) (void)hipGraphExecDestroy(p->graph_exec[i]);
    if (p->graph[i]) (void)hipGraphDestroy(p->graph[i]);
    p->graph_exec[i] = nullptr;
    p->graph[i] = nullptr;
    p->graph_sig[i].clear();
  }
}
