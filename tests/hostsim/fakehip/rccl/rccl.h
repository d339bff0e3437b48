/* TEST-ONLY stand-in for <rccl/rccl.h>: the types guber_global_sync.h names (the functions are looked up with dlsym at run time and
 * are never found in the CPU build of the engine).  Nothing in the product includes this file. */
#pragma once
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclUint = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
