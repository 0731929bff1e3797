// cpmppi_groups.hip — host code only: ENV GROUPS (include/cpmppi.h, cpmppi_groups_*).
//
// Independent MPPI problem instances need not march in step.  A launch of a few dozen envs (BASELINE configs[2] / configs[3]:
// one or two waves per SIMD) ends with its slowest wave - the waves of such a launch differ by 20-40 % in run time (rollouts at the
// track edge or spinning fast take the eventful path; profiles/r5/placement.txt) - and a chain step -> (plant ->) step serialises
// on it although only that wave's env depends on it.  The reference's only fan-out is share-nothing job arrays
// (others/EulerClusterScripts/ParallelDataGeneration.sh:2-17); the same structure inside one GPU: the E envs of a device as G
// contiguous groups, each with its own handle (a handle serves one stream) and its own stream, each running its own chain.  Group
// A's next step fills the SIMDs group B's slow waves leave idle, and a launch's serial tail hides under the other group's kernel.
//
// The streams come from cpmppi_stream_create: a hardware queue each.  The runtime multiplexes ordinary streams onto a handful of
// hardware queues, and two groups that land on one queue run one after the other (measured: C3 as two groups 327 us per step on
// pooled streams = serialised, 175 us on dedicated queues).
//
// cpmppi_groups_run enqueues K control periods of every group from C, round robin - two library calls per group and period, no
// interpreter in between; the argument blocks describe the FULL [E, ...] arrays and every group works on its slice in place.
#include <hip/hip_runtime.h>

#include <exception>
#include <string>
#include <vector>

#include "cpmppi.h"
#include "cpmppi_internal.hpp"

struct cpmppi_groups {
  struct Group {
    cpmppi_handle* h = nullptr;
    hipStream_t stream = nullptr;
    uint32_t first = 0, n = 0;
  };
  std::vector<Group> g;
  cpmppi_config cfg;
  int device = 0;
  uint32_t E = 0, env_offset = 0;
  hipEvent_t ev = nullptr;                 // fork / join
  std::string err;
};

namespace {

std::string g_groups_create_error;

int gfail(cpmppi_groups* g, int code, const std::string& msg) {
  if (g) g->err = msg; else g_groups_create_error = msg;
  return code;
}

template <typename T>
T* at(T* p, size_t off) { return p ? p + off : nullptr; }

size_t tiled_floats_per_env(const cpmppi_config& c) { return (size_t)((c.N + 63u) / 64u) * ((c.H + 3u) / 4u) * 256u; }

}  // namespace

extern "C" {

const char* cpmppi_groups_last_error(const cpmppi_groups* g) { return g ? g->err.c_str() : g_groups_create_error.c_str(); }

void cpmppi_groups_destroy(cpmppi_groups* g) {
  if (!g) return;
  int prev = -1;
  const bool sw = hipGetDevice(&prev) == hipSuccess && prev != g->device && hipSetDevice(g->device) == hipSuccess;
  for (auto& x : g->g) {
    if (x.stream) (void)hipStreamSynchronize(x.stream);
    if (x.h) cpmppi_destroy(x.h);
    if (x.stream) (void)cpmppi_stream_destroy(x.stream);
  }
  if (g->ev) (void)hipEventDestroy(g->ev);
  if (sw) (void)hipSetDevice(prev);
  delete g;
}

int cpmppi_groups_create(const cpmppi_config* cfg, int device, uint32_t groups, uint32_t env_offset, cpmppi_groups** out) {
  if (!cfg || !out) return gfail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_create: null argument");
  *out = nullptr;
  if (cfg->E == 0 || groups == 0) return gfail(nullptr, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_create: E and groups must be > 0");
  if (groups > cfg->E) groups = cfg->E;
  cpmppi_groups* g = nullptr;
  try {                                          // (no C++ exception leaves the C ABI)
  g = new cpmppi_groups();
  g->cfg = *cfg; g->device = device; g->E = cfg->E; g->env_offset = env_offset;
  const uint32_t base = cfg->E / groups, extra = cfg->E % groups;      // contiguous, as even as possible
  uint32_t e0 = 0;
  for (uint32_t i = 0; i < groups; ++i) {
    cpmppi_groups::Group x;
    x.first = e0; x.n = base + (i < extra ? 1u : 0u);
    cpmppi_config c = *cfg;
    c.E = x.n;
    int rc = cpmppi_create(&c, device, &x.h);
    if (rc != CPMPPI_OK) {
      const std::string msg = std::string("cpmppi_groups_create: ") + cpmppi_last_error(nullptr);
      cpmppi_groups_destroy(g);
      return gfail(nullptr, rc, msg);
    }
    void* st = nullptr;
    rc = cpmppi_stream_create(device, &st);
    x.stream = (hipStream_t)st;
    g->g.push_back(x);
    if (rc != CPMPPI_OK) {
      const std::string msg = std::string("cpmppi_groups_create: ") + cpmppi_last_error(nullptr);
      cpmppi_groups_destroy(g);
      return gfail(nullptr, rc, msg);
    }
    e0 += x.n;
  }
  int prev = -1;
  const bool sw = hipGetDevice(&prev) == hipSuccess && prev != device && hipSetDevice(device) == hipSuccess;
  const hipError_t e = hipEventCreateWithFlags(&g->ev, hipEventDisableTiming);
  if (sw) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    cpmppi_groups_destroy(g);
    return gfail(nullptr, CPMPPI_ERR_HIP, std::string("cpmppi_groups_create: hipEventCreate: ") + hipGetErrorString(e));
  }
  *out = g;
  return CPMPPI_OK;
  } catch (const std::exception&) {
    cpmppi_groups_destroy(g);
    try { g_groups_create_error = "cpmppi_groups_create: out of host memory"; } catch (...) {}
    return CPMPPI_ERR_NOMEM;
  }
}

uint32_t cpmppi_groups_count(const cpmppi_groups* g) { return g ? (uint32_t)g->g.size() : 0u; }

int cpmppi_groups_slice(const cpmppi_groups* g, uint32_t group, uint32_t* first_env, uint32_t* n_envs) {
  if (!g || group >= g->g.size()) return CPMPPI_ERR_BAD_ARG;
  if (first_env) *first_env = g->g[group].first;
  if (n_envs) *n_envs = g->g[group].n;
  return CPMPPI_OK;
}

cpmppi_handle* cpmppi_groups_handle(cpmppi_groups* g, uint32_t group) { return (g && group < g->g.size()) ? g->g[group].h : nullptr; }
void* cpmppi_groups_stream(cpmppi_groups* g, uint32_t group) { return (g && group < g->g.size()) ? (void*)g->g[group].stream : nullptr; }

int cpmppi_groups_fork(cpmppi_groups* g, void* stream) {
  if (!g) return CPMPPI_ERR_BAD_ARG;
  hipError_t e = hipEventRecord(g->ev, (hipStream_t)stream);
  for (auto& x : g->g)
    if (e == hipSuccess) e = hipStreamWaitEvent(x.stream, g->ev, 0);
  return e == hipSuccess ? CPMPPI_OK : gfail(g, CPMPPI_ERR_HIP, std::string("cpmppi_groups_fork: ") + hipGetErrorString(e));
}

int cpmppi_groups_join(cpmppi_groups* g, void* stream) {
  if (!g) return CPMPPI_ERR_BAD_ARG;
  hipError_t e = hipSuccess;
  for (auto& x : g->g) {
    if (e == hipSuccess) e = hipEventRecord(g->ev, x.stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, g->ev, 0);
  }
  return e == hipSuccess ? CPMPPI_OK : gfail(g, CPMPPI_ERR_HIP, std::string("cpmppi_groups_join: ") + hipGetErrorString(e));
}

}  // extern "C"

namespace {

// cpmppi_groups_run (recv_all == NULL) / cpmppi_groups_run_gather
int run_impl(cpmppi_groups* g, const cpmppi_step_args* step, const cpmppi_plant_args* plant, uint32_t periods, float* recv_all) {
  if (!g) return CPMPPI_ERR_BAD_ARG;
  cpmppi_comm::CommState* comm = nullptr;
  if (recv_all) {
    if (!step) return gfail(g, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_run_gather: a step argument block is required");
    comm = cpmppi_internal_comm(g->g[0].h);
    if (!comm) return gfail(g, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_run_gather: no communicator (cpmppi_groups_comm_init)");
    if (step->predictor == CPMPPI_PREDICTOR_GRU && step->u_nom_out && step->u_nom_out != step->u_nom)
      return gfail(g, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_run_gather: the GRU predictor steps in place");
  }
  if (!step && !plant) return gfail(g, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_run: neither a step nor a plant argument block");
  if (step && step->E != g->E) return gfail(g, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_run: step->E must be the groups' total env count");
  if (plant && plant->E != g->E) return gfail(g, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_run: plant->E must be the groups' total env count");
  if (step && step->offset_dev)
    return gfail(g, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_run: step->offset_dev is one shared device counter; the groups count their steps from step->offset");
  if (plant && plant->period_dev)
    return gfail(g, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_run: plant->period_dev is one shared device counter; the groups count their periods from plant->period");
  const cpmppi_config& c = g->cfg;
  const size_t N = c.N, H = c.H, P = (H + c.period - 1u) / c.period + 1u;
  try {
  // per-group argument blocks: the caller's, moved to the group's first env
  std::vector<cpmppi_step_args> sa(g->g.size());
  std::vector<cpmppi_plant_args> pa(g->g.size());
  for (size_t i = 0; i < g->g.size(); ++i) {
    const size_t e0 = g->g[i].first;
    if (step) {
      cpmppi_step_args a = *step;
      a.E = g->g[i].n;
      a.s0 = at(a.s0, e0 * 6); a.u_nom = at(a.u_nom, e0 * H); a.u_prev = at(a.u_prev, e0 * H); a.u_nom_out = at(a.u_nom_out, e0 * H);
      a.target_position = at(a.target_position, e0); a.target_equilibrium = at(a.target_equilibrium, e0); a.L = at(a.L, e0);
      const size_t per_env = a.noise_kind == CPMPPI_NOISE_DELTA_U ? N * H : (a.noise_kind == CPMPPI_NOISE_KNOTS ? N * P :
                             (a.noise_kind == CPMPPI_NOISE_DELTA_U_TILED ? tiled_floats_per_env(c) : 0));
      a.noise = at(a.noise, e0 * per_env);
      a.env_offset = step->env_offset + g->env_offset + (uint32_t)e0;
      a.Q_out = at(a.Q_out, e0); a.S_out = at(a.S_out, e0 * N); a.h0 = at(a.h0, e0 * 64); a.previous_input = at(a.previous_input, e0);
      sa[i] = a;
    }
    if (plant) {
      cpmppi_plant_args b = *plant;
      b.E = g->g[i].n;
      b.row_envs = plant->row_envs ? plant->row_envs : plant->E;
      b.s = at(b.s, e0 * 6); b.Q = at(b.Q, e0); b.L = at(b.L, e0);
      b.states_log = at(b.states_log, e0 * 6); b.dd_log = at(b.dd_log, e0 * 2); b.Q_log = at(b.Q_log, e0);
      b.target_position_table = at(b.target_position_table, e0); b.target_equilibrium_table = at(b.target_equilibrium_table, e0);
      b.L_table = at(b.L_table, e0);
      b.target_position_out = at(b.target_position_out, e0); b.target_equilibrium_out = at(b.target_equilibrium_out, e0);
      b.L_out = at(b.L_out, e0);
      b.m_pole = at(b.m_pole, e0); b.m_pole_table = at(b.m_pole_table, e0); b.L_controller_table = at(b.L_controller_table, e0);
      b.Q_disturbance_table = at(b.Q_disturbance_table, e0); b.Q_applied_out = at(b.Q_applied_out, e0);
      b.s_measured = at(b.s_measured, e0 * 6); b.state_history = at(b.state_history, e0 * 6);
      b.measurement_noise_table = at(b.measurement_noise_table, e0 * 4); b.angle_offset_table = at(b.angle_offset_table, e0);
      b.informed_table = at(b.informed_table, e0);
      pa[i] = b;
    }
  }
  const bool alternate = comm && step->u_nom_out && step->u_nom_out != step->u_nom;
  for (uint32_t k = 0; k < periods; ++k) {
    cpmppi_comm::GatherTicket ticket{};
    float* out_all = nullptr;
    if (comm) {
      // a device-side wait gave up (a peer stalled beyond the timeout): say so now (as cpmppi_step_gather does)
      if (cpmppi_comm::comm_error_pending(g->g[0].h))
        return gfail(g, CPMPPI_ERR_COMM, "cpmppi_groups_run_gather: an earlier step's device-side wait for an all-gather timed out; "
                                         "cpmppi_comm_sync(cpmppi_groups_handle(g, 0)) reports and clears the condition");
      // ONE ticket per period, shared by the launches of every group: the step number all envs of the device publish together
      const bool swapped = alternate && (k & 1u);
      out_all = alternate ? (swapped ? step->u_nom : step->u_nom_out) : step->u_nom;
      cpmppi_comm::begin_step_gather(comm, out_all, &ticket);
      ticket.envs = g->E;
      if (alternate)
        for (size_t i = 0; i < g->g.size(); ++i) {
          const size_t off = (size_t)g->g[i].first * H;
          sa[i].u_nom = (swapped ? step->u_nom_out : step->u_nom) + off;
          sa[i].u_nom_out = out_all + off;
        }
    }
    for (size_t i = 0; i < g->g.size(); ++i) {
      cpmppi_handle* h = g->g[i].h;
      if (step) {
        sa[i].offset = step->offset + k;
        if (comm) {
          const int rg = cpmppi_comm::enqueue_guard(g->g[0].h, ticket, g->E, g->g[i].stream);
          if (rg != CPMPPI_OK) return gfail(g, rg, std::string("cpmppi_groups_run_gather: ") + cpmppi_last_error(g->g[0].h));
        }
        const int rc = comm ? cpmppi_internal_step_ticket(h, &sa[i], g->g[i].stream, &ticket) : cpmppi_step(h, &sa[i], g->g[i].stream);
        if (rc != CPMPPI_OK) {
          if (comm && i > 0) cpmppi_comm::poison(comm);      // other groups' launches of this period are out: their arrivals will never be complete
          return gfail(g, rc, std::string("cpmppi_groups_run: group ") + std::to_string(i) + ": " + cpmppi_last_error(h));
        }
      }
      if (plant) {
        pa[i].period = plant->period + k;
        const int rc = cpmppi_plant_step(h, &pa[i], g->g[i].stream);
        if (rc != CPMPPI_OK) return gfail(g, rc, std::string("cpmppi_groups_run: group ") + std::to_string(i) + ": " + cpmppi_last_error(h));
      }
    }
    if (comm) {
      // side stream: wait until the LAST env of the LAST group has published this step -> all-gather of the device's whole
      // u_nom[E, H] -> post its completion.  (A launch of this period that failed above has returned already: the side stream
      // is then left without this period's wait; cpmppi_comm_sync's escape covers a step that never publishes.)
      const int rc = cpmppi_comm::enqueue_gather(g->g[0].h, out_all, recv_all, (size_t)g->E * H);
      if (rc != CPMPPI_OK) return gfail(g, rc, std::string("cpmppi_groups_run_gather: ") + cpmppi_last_error(g->g[0].h));
    }
  }
  return CPMPPI_OK;
  } catch (const std::exception&) {
    return CPMPPI_ERR_NOMEM;
  }
}

}  // namespace

extern "C" {

int cpmppi_groups_run(cpmppi_groups* g, const cpmppi_step_args* step, const cpmppi_plant_args* plant, uint32_t periods) {
  return run_impl(g, step, plant, periods, nullptr);
}

int cpmppi_groups_run_gather(cpmppi_groups* g, const cpmppi_step_args* step, const cpmppi_plant_args* plant, uint32_t periods,
                             float* recv_all) {
  if (!g) return CPMPPI_ERR_BAD_ARG;
  if (!recv_all) return gfail(g, CPMPPI_ERR_BAD_ARG, "cpmppi_groups_run_gather: recv_all is required");
  return run_impl(g, step, plant, periods, recv_all);
}

// ONE communicator and ONE side stream for all env groups of the device; it lives in group 0's handle (cpmppi_comm_* calls take
// cpmppi_groups_handle(g, 0)) and goes with it in cpmppi_groups_destroy.
int cpmppi_groups_comm_init(cpmppi_groups* g, const void* id, int world, int rank, const char* rccl_path) {
  if (!g || g->g.empty()) return CPMPPI_ERR_BAD_ARG;
  const int rc = cpmppi_comm_init(g->g[0].h, id, world, rank, rccl_path);
  if (rc != CPMPPI_OK) return gfail(g, rc, std::string("cpmppi_groups_comm_init: ") + cpmppi_last_error(g->g[0].h));
  if (cpmppi_comm::share_between_groups(cpmppi_internal_comm(g->g[0].h)) != CPMPPI_OK) {
    (void)cpmppi_comm_destroy(g->g[0].h);
    return gfail(g, CPMPPI_ERR_HIP, "cpmppi_groups_comm_init: could not switch the communicator to the env-group form");
  }
  return CPMPPI_OK;
}

}  // extern "C"
