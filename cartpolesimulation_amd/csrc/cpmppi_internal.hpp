// cpmppi_internal.hpp — what the translation units of libcpmppi.so share beside the public header: access to the few
// handle fields the communicator unit (cpmppi_comm.hip) needs.  The handle's layout stays private to cpmppi.hip.
#pragma once
#include <string>
#include "cpmppi.h"

namespace cpmppi_comm {
struct CommState;                                        // cpmppi_comm.hip
void destroy(CommState* c);                              // called by cpmppi_destroy

// cpmppi_step_gather = begin_step_gather (numbers the gather, says what the step's finalize must publish / await) ->
// the step launch -> enqueue_gather (side stream: wait for the published step, ncclAllGather, post completion)
struct GatherTicket {
  unsigned* flags;        // device words: [0] envs finalized, [1] steps published, [2] gathers completed, [3] error, [4..9] the
                          // slow paths' pointers and the timeout (GatherSync, cpmppi_rollout.hpp)
  unsigned publish, need;
  unsigned envs;          // envs that publish the step together: 0 = the launch's own (one handle); env groups: all groups' envs
};
void begin_step_gather(CommState* c, const float* out_buffer, GatherTicket* out);
int share_between_groups(CommState* c);      // the communicator serves env groups: steps alternate between two flag blocks
void abort_step_gather(CommState* c);
int enqueue_gather(cpmppi_handle* h, const float* send, float* recv_all, size_t count);
int enqueue_guard(cpmppi_handle* h, const GatherTicket& t, unsigned envs, void* stream);   // launch stream: gather_guard_kernel (many envs only)
void poison(CommState* c);                   // a partly enqueued step-gather: error state until cpmppi_comm_sync
int comm_error_pending(cpmppi_handle* h);    // a device-side wait of this handle has timed out (sticky until cpmppi_comm_sync)
}  // namespace cpmppi_comm

cpmppi_comm::CommState*& cpmppi_internal_comm(cpmppi_handle* h);
// cpmppi_step whose finalize takes part in a step-gather described by `ticket` (cpmppi_step_gather; cpmppi_groups_run_gather, where
// the ticket is shared by the launches of every group)
int cpmppi_internal_step_ticket(cpmppi_handle* h, const cpmppi_step_args* a, void* stream, const cpmppi_comm::GatherTicket* ticket);
int cpmppi_internal_device(const cpmppi_handle* h);
int cpmppi_internal_fail(cpmppi_handle* h, int code, const std::string& msg);
