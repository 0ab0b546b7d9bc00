// tests/hostsim/rccl/rccl.h -- TEST INFRASTRUCTURE: the RCCL types og_cluster.inl names (the library binds RCCL with dlopen,
// and only when a cluster spans more than one device -- which the host simulator never reports).
#pragma once
#include <hip/hip_runtime.h>
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
