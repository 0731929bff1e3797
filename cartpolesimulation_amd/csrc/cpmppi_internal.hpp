// cpmppi_internal.hpp — what the translation units of libcpmppi.so share beside the public header: access to the few
// handle fields the communicator unit (cpmppi_comm.hip) needs.  The handle's layout stays private to cpmppi.hip.
#pragma once
#include <string>
#include "cpmppi.h"

namespace cpmppi_comm {
struct CommState;                                        // cpmppi_comm.hip
void destroy(CommState* c);                              // called by cpmppi_destroy
}  // namespace cpmppi_comm

cpmppi_comm::CommState*& cpmppi_internal_comm(cpmppi_handle* h);
int cpmppi_internal_device(const cpmppi_handle* h);
int cpmppi_internal_fail(cpmppi_handle* h, int code, const std::string& msg);
