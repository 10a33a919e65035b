// Stand-in for <rccl/rccl.h> under the host emulation of the tests: the handful of entry points the library looks up
// at run time, served by tests/emu/emu_rccl.cpp instead of a shared library (dlopen/dlsym are redirected below; the
// library includes <dlfcn.h> before this header).
#pragma once
#include <hip/hip_runtime.h>

typedef struct emu_comm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclHalf = 6,
               ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;

extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}

void* emu_dlopen(const char* name, int flags);
void* emu_dlsym(void* handle, const char* symbol);
#define dlopen(name, flags) emu_dlopen((name), (flags))
#define dlsym(handle, symbol) emu_dlsym((handle), (symbol))
