// cvo_comm.h -- RCCL glue: the two tiny all-reduces per align() iteration
// when the target cloud is sharded over the GPUs of a node.
// RCCL is bound lazily with dlopen so that the single-GPU path has no RCCL
// dependency at load time and, inside a PyTorch process, shares the RCCL
// instance torch already loaded.
#pragma once

#include <hip/hip_runtime.h>

struct cvo_comm;

// 0 on success
int cvo_comm_unique_id(void *id_bytes_128);
cvo_comm *cvo_comm_create(const void *id_bytes_128, int rank, int world);
void cvo_comm_destroy(cvo_comm *c);
// in-place float64 sum over ranks, ordered on `stream`
int cvo_comm_allreduce(cvo_comm *c, double *dev_buf, int count, hipStream_t stream);
const char *cvo_comm_last_error(const cvo_comm *c);
