/* A C consumer of libcpmppi.so: no torch, no C++ - the HIP runtime's C API for the device buffers and include/cpmppi.h.
 * tests/test_gpu_c_consumer.py builds it with gcc, runs it and compares its output with the Python binding's.
 *   consumer <config.bin> <E> <seed> <steps>
 * Reads a cpmppi_config blob (written by the test from the ctypes mirror), runs `steps` fused MPPI steps for E envs
 * from fixed states with in-kernel Philox noise through cpmppi_step (device pointers) and again through cpmppi_step_host
 * (host pointers), prints Q of every env per step ("Q ..." / "Qh ...") and the final nominal sequence of env 0; then the
 * same steps a third time through the multi-GPU entry points with ONE rank (cpmppi_comm_unique_id / cpmppi_comm_init /
 * cpmppi_step_gather with two alternating nominal-sequence buffers / cpmppi_comm_sync): "Qg ..." per step, and "g ..." =
 * the gathered copy of env 0's final sequence (must equal "u ...").  Last, the data generator's device loop through the env-group
 * entry points (cpmppi_groups_create / fork / run / join) with cpmppi_plant_step's schedule tables and recording: "R", "Qc", "D". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include "cpmppi.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: consumer config.bin E seed steps\n"); return 1; }
  cpmppi_config cfg;
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(&cfg, sizeof cfg, 1, f) != 1) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  fclose(f);
  const uint32_t E = (uint32_t)atoi(argv[2]);
  const uint64_t seed = (uint64_t)atoll(argv[3]);
  const int steps = atoi(argv[4]);
  const uint32_t H = cfg.H;
  cpmppi_handle* h = NULL;
  if (cpmppi_create(&cfg, 0, &h) != CPMPPI_OK) { fprintf(stderr, "create: %s\n", cpmppi_last_error(NULL)); return 3; }
  printf("%s\n", cpmppi_version());

  float* hs = (float*)calloc((size_t)E * 6, sizeof(float));
  float* ht = (float*)calloc((size_t)E * 3, sizeof(float));
  for (uint32_t e = 0; e < E; ++e) {              /* angle, angleD, cos, sin, position, positionD */
    const float th = 0.05f + 0.1f * (float)e;
    hs[6 * e + 0] = th; hs[6 * e + 1] = -0.2f * (float)e; hs[6 * e + 2] = 1.0f - 0.5f * th * th; hs[6 * e + 3] = th - th * th * th / 6.0f;
    hs[6 * e + 4] = 0.01f * (float)e; hs[6 * e + 5] = 0.0f;
    ht[e] = 0.02f; ht[E + e] = 1.0f; ht[2 * E + e] = 0.395f;        /* target_position, target_equilibrium, L */
  }
  float *s0, *tgt, *u_nom, *Q;
  HIPCHECK(hipMalloc((void**)&s0, (size_t)E * 6 * sizeof(float)));
  HIPCHECK(hipMalloc((void**)&tgt, (size_t)E * 3 * sizeof(float)));
  HIPCHECK(hipMalloc((void**)&u_nom, (size_t)E * H * sizeof(float)));
  HIPCHECK(hipMalloc((void**)&Q, (size_t)E * sizeof(float)));
  HIPCHECK(hipMemcpy(s0, hs, (size_t)E * 6 * sizeof(float), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(tgt, ht, (size_t)E * 3 * sizeof(float), hipMemcpyHostToDevice));
  HIPCHECK(hipMemset(u_nom, 0, (size_t)E * H * sizeof(float)));

  float* hq = (float*)malloc((size_t)E * sizeof(float));
  for (int it = 0; it < steps; ++it) {
    cpmppi_step_args a;
    memset(&a, 0, sizeof a);
    a.E = E; a.s0 = s0; a.u_nom = u_nom; a.target_position = tgt; a.target_equilibrium = tgt + E; a.L = tgt + 2 * E;
    a.noise_kind = CPMPPI_NOISE_PHILOX; a.seed = seed; a.offset = (uint64_t)it; a.Q_out = Q;
    if (cpmppi_step(h, &a, NULL) != CPMPPI_OK) { fprintf(stderr, "step: %s\n", cpmppi_last_error(h)); return 4; }
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(hq, Q, (size_t)E * sizeof(float), hipMemcpyDeviceToHost));
    printf("Q");
    for (uint32_t e = 0; e < E; ++e) printf(" %.9g", hq[e]);
    printf("\n");
  }
  /* the same steps again through the host-pointer entry point: state, targets and Q never leave host memory on this side */
  HIPCHECK(hipMemset(u_nom, 0, (size_t)E * H * sizeof(float)));
  for (int it = 0; it < steps; ++it) {
    if (cpmppi_step_host(h, E, hs, ht, ht + E, ht + 2 * E, u_nom, seed, (uint64_t)it, 0, hq, NULL) != CPMPPI_OK) {
      fprintf(stderr, "step_host: %s\n", cpmppi_last_error(h));
      return 5;
    }
    printf("Qh");
    for (uint32_t e = 0; e < E; ++e) printf(" %.9g", hq[e]);
    printf("\n");
  }
  float* hu = (float*)malloc((size_t)H * sizeof(float));
  HIPCHECK(hipMemcpy(hu, u_nom, (size_t)H * sizeof(float), hipMemcpyDeviceToHost));
  printf("u");
  for (uint32_t k = 0; k < H; ++k) printf(" %.9g", hu[k]);
  printf("\n");
  /* ---- one rank of the env-sharded form: step + all-gather of its result in one call ---- */
  {
    unsigned char id[CPMPPI_COMM_ID_BYTES];
    float *ub[2], *recv[2];
    if (cpmppi_comm_unique_id(id, NULL) != CPMPPI_OK) { fprintf(stderr, "comm id: %s\n", cpmppi_last_error(NULL)); return 6; }
    if (cpmppi_comm_init(h, id, 1, 0, NULL) != CPMPPI_OK) { fprintf(stderr, "comm init: %s\n", cpmppi_last_error(h)); return 6; }
    /* stamped blocks: CPMPPI_GATHER_STAMP_FLOATS words behind the sequences of every buffer and of every rank's block */
    if (cpmppi_comm_set_stamped(h, 1) != CPMPPI_OK) { fprintf(stderr, "stamped: %s\n", cpmppi_last_error(h)); return 6; }
    const size_t nblk = (size_t)E * H + CPMPPI_GATHER_STAMP_FLOATS;
    for (int b = 0; b < 2; ++b) {
      HIPCHECK(hipMalloc((void**)&ub[b], nblk * sizeof(float)));
      HIPCHECK(hipMalloc((void**)&recv[b], nblk * sizeof(float)));
      HIPCHECK(hipMemset(ub[b], 0, nblk * sizeof(float)));
      HIPCHECK(hipMemset(recv[b], 0, nblk * sizeof(float)));
    }
    for (int it = 0; it < steps; ++it) {
      cpmppi_step_args a;
      memset(&a, 0, sizeof a);
      a.E = E; a.s0 = s0; a.u_nom = ub[it & 1]; a.u_nom_out = ub[(it + 1) & 1];
      a.target_position = tgt; a.target_equilibrium = tgt + E; a.L = tgt + 2 * E;
      a.noise_kind = CPMPPI_NOISE_PHILOX; a.seed = seed; a.offset = (uint64_t)it; a.Q_out = Q;
      if (cpmppi_step_gather(h, &a, recv[(it + 1) & 1], NULL) != CPMPPI_OK) { fprintf(stderr, "step_gather: %s\n", cpmppi_last_error(h)); return 7; }
      HIPCHECK(hipDeviceSynchronize());
      HIPCHECK(hipMemcpy(hq, Q, (size_t)E * sizeof(float), hipMemcpyDeviceToHost));
      printf("Qg");
      for (uint32_t e = 0; e < E; ++e) printf(" %.9g", hq[e]);
      printf("\n");
    }
    if (cpmppi_comm_sync(h) != CPMPPI_OK) { fprintf(stderr, "comm sync: %s\n", cpmppi_last_error(h)); return 8; }
    HIPCHECK(hipMemcpy(hu, recv[steps & 1], (size_t)H * sizeof(float), hipMemcpyDeviceToHost));
    printf("g");
    for (uint32_t k = 0; k < H; ++k) printf(" %.9g", hu[k]);
    printf("\n");
    uint32_t stamp = 0;                                     /* the number of the step-gather that produced the block */
    HIPCHECK(hipMemcpy(&stamp, recv[steps & 1] + (size_t)E * H, sizeof stamp, hipMemcpyDeviceToHost));
    printf("st %u\n", stamp);
    cpmppi_comm_destroy(h);
  }
  cpmppi_destroy(h);
  /* ---- the data generator's device loop from plain C: env groups (two handles, two dedicated-queue streams), a moving target
   * and a flipping equilibrium read from schedule tables, rows saved every 5 simulation steps - T control periods of both groups
   * enqueued by ONE cpmppi_groups_run call, then the run's last controller call.  "R ..." = the last saved state of every env,
   * "Qc ..." = the last control: must equal the Python harness (one chain, no groups) bit for bit. ---- */
  {
    const uint32_t T = (uint32_t)steps + 3, n_ctrl = 10, n_save = 5, stride = 5, rows = T * n_ctrl / n_save + 1, srows = T * n_ctrl / stride + 1;
    cpmppi_groups* gr = NULL;
    if (cpmppi_abi_version() != CPMPPI_ABI_VERSION) { fprintf(stderr, "abi %u\n", cpmppi_abi_version()); return 9; }
    if (cpmppi_groups_create(&cfg, 0, 2, 0, &gr) != CPMPPI_OK) { fprintf(stderr, "groups: %s\n", cpmppi_groups_last_error(NULL)); return 9; }
    float* htab = (float*)malloc((size_t)srows * E * 2 * sizeof(float));
    for (uint32_t r = 0; r < srows; ++r)
      for (uint32_t e = 0; e < E; ++e) {
        htab[(size_t)r * E + e] = 0.002f * (float)((r * 7u + e * 3u) % 40u) - 0.04f;                    /* target position */
        htab[(size_t)(srows + r) * E + e] = ((r / 6u + e) % 2u) ? -1.0f : 1.0f;                          /* target equilibrium */
      }
    float *s, *un, *Qd, *tab, *cur, *slog, *ddlog, *Qlog;
    HIPCHECK(hipMalloc((void**)&s, (size_t)E * 6 * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&un, (size_t)E * H * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&Qd, (size_t)E * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&tab, (size_t)srows * E * 2 * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&cur, (size_t)E * 2 * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&slog, (size_t)rows * E * 6 * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&ddlog, (size_t)rows * E * 2 * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&Qlog, (size_t)(T + 1) * E * sizeof(float)));
    HIPCHECK(hipMemcpy(s, hs, (size_t)E * 6 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(slog, hs, (size_t)E * 6 * sizeof(float), hipMemcpyHostToDevice));                /* row 0: the initial state */
    HIPCHECK(hipMemset(un, 0, (size_t)E * H * sizeof(float)));
    HIPCHECK(hipMemcpy(tab, htab, (size_t)srows * E * 2 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(cur, htab, (size_t)E * sizeof(float), hipMemcpyHostToDevice));                     /* row 0 of both tables */
    HIPCHECK(hipMemcpy(cur + E, htab + (size_t)srows * E, (size_t)E * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(hipDeviceSynchronize());
    cpmppi_step_args a;
    memset(&a, 0, sizeof a);
    a.E = E; a.s0 = s; a.u_nom = un; a.target_position = cur; a.target_equilibrium = cur + E; a.L = tgt + 2 * E;
    a.noise_kind = CPMPPI_NOISE_PHILOX; a.seed = seed; a.offset = 0; a.Q_out = Qd;
    cpmppi_plant_args b;
    memset(&b, 0, sizeof b);
    b.E = E; b.s = s; b.Q = Qd; b.L = tgt + 2 * E; b.n_substeps = n_ctrl; b.period_steps = n_ctrl; b.dt_sim = 0.002f; b.period = 0;
    b.states_log = slog; b.dd_log = ddlog; b.save_rows = rows; b.save_every = n_save; b.Q_log = Qlog; b.ctrl_rows = T + 1;
    b.target_position_table = tab; b.target_equilibrium_table = tab + (size_t)srows * E; b.sched_rows = srows; b.sched_stride = stride;
    b.target_position_out = cur; b.target_equilibrium_out = cur + E;
    /* ... with the groups under ONE communicator (one rank here) and one all-gather of the device's u_nom[E, H] per period:
     * cpmppi_groups_comm_init + cpmppi_groups_run_gather instead of cpmppi_groups_run - the recording must not change by a bit */
    unsigned char gid[CPMPPI_COMM_ID_BYTES];
    float* grecv;
    HIPCHECK(hipMalloc((void**)&grecv, ((size_t)E * H + CPMPPI_GATHER_STAMP_FLOATS) * sizeof(float)));
    if (cpmppi_comm_unique_id(gid, NULL) != CPMPPI_OK || cpmppi_groups_comm_init(gr, gid, 1, 0, NULL) != CPMPPI_OK) { fprintf(stderr, "groups comm: %s\n", cpmppi_groups_last_error(gr)); return 9; }
    if (cpmppi_groups_fork(gr, NULL) != CPMPPI_OK || cpmppi_groups_run_gather(gr, &a, &b, T, grecv) != CPMPPI_OK) { fprintf(stderr, "groups run: %s\n", cpmppi_groups_last_error(gr)); return 9; }
    a.offset = T; b.period = T; b.n_substeps = 0;                                                         /* the last controller call: record only */
    if (cpmppi_groups_run_gather(gr, &a, &b, 1, grecv) != CPMPPI_OK || cpmppi_groups_join(gr, NULL) != CPMPPI_OK) { fprintf(stderr, "groups run: %s\n", cpmppi_groups_last_error(gr)); return 9; }
    HIPCHECK(hipDeviceSynchronize());
    if (cpmppi_comm_sync(cpmppi_groups_handle(gr, 0)) != CPMPPI_OK) { fprintf(stderr, "groups comm sync: %s\n", cpmppi_last_error(cpmppi_groups_handle(gr, 0))); return 9; }
    {
      cpmppi_comm_info info;
      float* hg = (float*)malloc((size_t)E * H * sizeof(float));
      float* hn = (float*)malloc((size_t)E * H * sizeof(float));
      if (cpmppi_comm_get_info(cpmppi_groups_handle(gr, 0), &info) != CPMPPI_OK) return 9;
      HIPCHECK(hipMemcpy(hg, grecv, (size_t)E * H * sizeof(float), hipMemcpyDeviceToHost));
      HIPCHECK(hipMemcpy(hn, un, (size_t)E * H * sizeof(float), hipMemcpyDeviceToHost));
      printf("gg %d %u %d\n", memcmp(hg, hn, (size_t)E * H * sizeof(float)) == 0, info.gathers_enqueued, info.rccl_ranks);   /* the last gather = u_nom */
      free(hg); free(hn);
    }
    float* hr = (float*)malloc((size_t)E * 6 * sizeof(float));
    HIPCHECK(hipMemcpy(hr, slog + (size_t)(rows - 1) * E * 6, (size_t)E * 6 * sizeof(float), hipMemcpyDeviceToHost));
    printf("R");
    for (uint32_t i = 0; i < E * 6; ++i) printf(" %.9g", hr[i]);
    printf("\n");
    HIPCHECK(hipMemcpy(hq, Qlog + (size_t)T * E, (size_t)E * sizeof(float), hipMemcpyDeviceToHost));
    printf("Qc");
    for (uint32_t e = 0; e < E; ++e) printf(" %.9g", hq[e]);
    printf("\n");
    HIPCHECK(hipMemcpy(hr, ddlog + (size_t)(rows - 1) * E * 2, (size_t)E * 2 * sizeof(float), hipMemcpyDeviceToHost));
    printf("D");
    for (uint32_t i = 0; i < E * 2; ++i) printf(" %.9g", hr[i]);
    printf("\n");
    cpmppi_groups_destroy(gr);
  }
  return 0;
}
