// Stub of the RCCL types the host code names (the library loads librccl.so with dlopen: no symbol is linked).
#pragma once
#include <cstddef>
typedef struct tsan_nccl_comm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1 } ncclResult_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
