/*
 * rt_comm.h -- the exchange step of a multi-GPU render: one process per GPU, the tile buffers of all ranks sent to
 * one of them with RCCL point-to-point calls over xGMI.  RCCL is opened at run time (dlopen) by the first call, so a
 * process that renders on one GPU never loads it and the library does not link against it.
 */
#ifndef RT_COMM_H
#define RT_COMM_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace rtc {

/* 128 bytes that identify a communicator: made by one process, handed to the others by any means (file, pipe, MPI) */
bool unique_id(char out[128], std::string& err);
/* collective over the `world` processes that hold the same id; the current HIP device is this rank's GPU */
bool init(const char id[128], int rank, int world, std::string& err);
void destroy();
bool active();
int rank();
int world();
/* Every rank contributes `count` floats at d_mine; on `root` they arrive rank-major in d_gathered (world * count
 * floats, the root's own part copied on the stream); elsewhere d_gathered is not used.  Enqueued on `st`. */
bool gather(const float* d_mine, size_t count, float* d_gathered, int root, hipStream_t st, std::string& err);

}  // namespace rtc

#endif
