// Peer-mapped buffers, interprocess events and copy-engine pushes: what the "sdma" transport of the sharded step
// (scone_amd/distributed.py, gather_transport="sdma") is built from.  The exchange of the all-gather form sends EXACT
// contiguous ranges (a rank's payload rows, scales and hash fragment go to the same offsets of every peer's receive
// buffers), so it needs no kernel at all: every rank maps its peers' receive buffers once (hipIpcGetMemHandle /
// hipIpcOpenMemHandle), pushes its three columns with hipMemcpyAsync to the peer pointers -- the copy engines (SDMA) move
// them over xGMI while every wave slot of the chip belongs to the lookup kernel -- and says "my pushes for this slot are
// complete" with an interprocess event the receivers wait for.  RCCL's send / recv are kernels: a few workgroups per peer
// that must find room beside a lookup grid that fills the chip (tools/c5_rank0_step.py --transport-standin prices that).
// New here: the reference is one process with no table exchange at all (hydra_train.py:32-48 is its only collective set-up).
#include "scone_common.h"

#include <cstring>

extern "C" int scone_ipc_alloc(scone_handle *h, uint64_t bytes, void **d_ptr, void *handle64) {
  if (!h || !d_ptr || !handle64 || bytes == 0) return h ? scone_fail(h, SCONE_EINVAL, "scone_ipc_alloc: bad argument") : SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  *d_ptr = nullptr;
  void *p = nullptr;
  SCONE_HIP(h, hipMalloc(&p, (size_t)bytes));
  hipIpcMemHandle_t mh;
  hipError_t e = hipIpcGetMemHandle(&mh, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return scone_hip_fail(h, e, "hipIpcGetMemHandle");
  }
  static_assert(sizeof(mh) == 64, "hipIpcMemHandle_t is 64 bytes");
  memcpy(handle64, &mh, sizeof(mh));
  *d_ptr = p;
  return SCONE_OK;
}

extern "C" int scone_ipc_free(scone_handle *h, void *d_ptr) {
  if (!h) return SCONE_EINVAL;
  if (!d_ptr) return SCONE_OK;
  SCONE_ON_DEVICE(h);
  SCONE_HIP(h, hipFree(d_ptr));
  return SCONE_OK;
}

extern "C" int scone_ipc_open(scone_handle *h, const void *handle64, void **d_ptr) {
  if (!h || !handle64 || !d_ptr) return h ? scone_fail(h, SCONE_EINVAL, "scone_ipc_open: bad argument") : SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  hipIpcMemHandle_t mh;
  memcpy(&mh, handle64, sizeof(mh));
  *d_ptr = nullptr;
  SCONE_HIP(h, hipIpcOpenMemHandle(d_ptr, mh, hipIpcMemLazyEnablePeerAccess));
  return SCONE_OK;
}

extern "C" int scone_ipc_close(scone_handle *h, void *d_ptr) {
  if (!h) return SCONE_EINVAL;
  if (!d_ptr) return SCONE_OK;
  SCONE_ON_DEVICE(h);
  SCONE_HIP(h, hipIpcCloseMemHandle(d_ptr));
  return SCONE_OK;
}

extern "C" int scone_ipc_event_create(scone_handle *h, void **event, void *handle64) {
  if (!h || !event || !handle64) return h ? scone_fail(h, SCONE_EINVAL, "scone_ipc_event_create: bad argument") : SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  hipEvent_t ev = nullptr;
  SCONE_HIP(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess));
  hipIpcEventHandle_t eh;
  hipError_t e = hipIpcGetEventHandle(&eh, ev);
  if (e != hipSuccess) {
    (void)hipEventDestroy(ev);
    return scone_hip_fail(h, e, "hipIpcGetEventHandle");
  }
  static_assert(sizeof(eh) == 64, "hipIpcEventHandle_t is 64 bytes");
  memcpy(handle64, &eh, sizeof(eh));
  *event = ev;
  return SCONE_OK;
}

extern "C" int scone_ipc_event_open(scone_handle *h, const void *handle64, void **event) {
  if (!h || !handle64 || !event) return h ? scone_fail(h, SCONE_EINVAL, "scone_ipc_event_open: bad argument") : SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  hipIpcEventHandle_t eh;
  memcpy(&eh, handle64, sizeof(eh));
  hipEvent_t ev = nullptr;
  SCONE_HIP(h, hipIpcOpenEventHandle(&ev, eh));
  *event = ev;
  return SCONE_OK;
}

extern "C" int scone_ipc_event_destroy(scone_handle *h, void *event) {
  if (!h) return SCONE_EINVAL;
  if (!event) return SCONE_OK;
  SCONE_ON_DEVICE(h);
  SCONE_HIP(h, hipEventDestroy((hipEvent_t)event));
  return SCONE_OK;
}

extern "C" int scone_ipc_event_record(scone_handle *h, void *event, scone_stream_t stream) {
  if (!h || !event) return h ? scone_fail(h, SCONE_EINVAL, "scone_ipc_event_record: null event") : SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  SCONE_HIP(h, hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return SCONE_OK;
}

extern "C" int scone_ipc_event_wait(scone_handle *h, void *event, scone_stream_t stream) {
  if (!h || !event) return h ? scone_fail(h, SCONE_EINVAL, "scone_ipc_event_wait: null event") : SCONE_EINVAL;
  SCONE_ON_DEVICE(h);
  SCONE_HIP(h, hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return SCONE_OK;
}

// bytes from this device to a (peer-mapped) device pointer.  copy_engine != 0: hipMemcpyDeviceToDeviceNoCU -- the runtime must
// not fall back to a blit kernel (same-device copies and some topologies would otherwise use one)
extern "C" int scone_ipc_push(scone_handle *h, void *d_dst, const void *d_src, uint64_t bytes, int32_t copy_engine,
                              scone_stream_t stream) {
  if (!h) return SCONE_EINVAL;
  if (bytes == 0) return SCONE_OK;
  if (!d_dst || !d_src) return scone_fail(h, SCONE_EINVAL, "scone_ipc_push: null pointer");
  SCONE_ON_DEVICE(h);
  SCONE_HIP(h, hipMemcpyAsync(d_dst, d_src, (size_t)bytes, copy_engine ? hipMemcpyDeviceToDeviceNoCU : hipMemcpyDeviceToDevice,
                              (hipStream_t)stream));
  return SCONE_OK;
}
