// clora_comm.hip -- the data-parallel exchange of the path behind the C ABI: ONE in-place all-reduce (sum) of the flat fp32
// adapter-gradient buffer per optimizer step, RCCL over xGMI (SURVEY.md section 8b "allreduce_flat", 8e).
//
// Reference: the gradient all-reduce that accelerate's DDP wrapper performs inside `accelerator.backward(loss)` for the prepared
// `control_lora` (reference train_text_to_image_control_lora.py:683-685, 790).  Here the 6.05 M trainable values live in one
// flat buffer (controllora_amd/train.py FlatParams), so the exchange is a single 24.19 MB collective; the 1/N of the mean is
// folded into the optimizer kernels (clora_optim_prep_f32).
//
// One process per GPU; the communicator is the one per-process handle the ABI keeps (besides the tuning knobs): it is created
// from a 128-byte unique id that rank 0 obtains with clora_comm_unique_id and the host shares by any means it likes
// (torch.distributed broadcast in controllora_amd/train.py; a file or socket from a C host).
//
// librccl is opened lazily with dlopen: libclora.so carries no link-time dependency on it, single-GPU users never load it, and
// a host without RCCL gets CLORA_ERR_LAUNCH from the clora_comm_* calls instead of a loader error.
#include <dlfcn.h>
#include <link.h>
#include <string.h>
#include "clora_common.h"
#include "../../include/clora.h"

namespace {

typedef struct { char internal[128]; } nccl_uid_t;          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm_t;
typedef int (*get_uid_fn)(nccl_uid_t*);
typedef int (*init_rank_fn)(nccl_comm_t*, int, nccl_uid_t, int);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*destroy_fn)(nccl_comm_t);

struct Rccl {
    void* handle = nullptr;
    get_uid_fn get_uid = nullptr;
    init_rank_fn init_rank = nullptr;
    all_reduce_fn all_reduce = nullptr;
    destroy_fn destroy = nullptr;
    nccl_comm_t comm = nullptr;
    int world = 0, rank = -1;
};
Rccl g_rccl;

// A librccl that the process has ALREADY mapped (a torch process: torch/lib/librccl.so, soname librccl.so.1) must be the one this
// library binds -- two RCCL instances in one process would each own their own topology / IPC state.  dlopen by soname finds it
// only if it was loaded under that soname; walking the loaded objects finds it under any path.
int find_mapped_rccl(struct dl_phdr_info* info, size_t, void* out) {
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) { strncpy((char*)out, info->dlpi_name, 1023); return 1; }
    return 0;
}

bool rccl_load() {
    if (g_rccl.handle) return true;
    void* h = nullptr;
    char mapped[1024] = {0};
    if (dl_iterate_phdr(find_mapped_rccl, mapped) && mapped[0]) h = dlopen(mapped, RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    g_rccl.get_uid = (get_uid_fn)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (init_rank_fn)dlsym(h, "ncclCommInitRank");
    g_rccl.all_reduce = (all_reduce_fn)dlsym(h, "ncclAllReduce");
    g_rccl.destroy = (destroy_fn)dlsym(h, "ncclCommDestroy");
    if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.all_reduce || !g_rccl.destroy) { dlclose(h); return false; }
    g_rccl.handle = h;
    return true;
}

constexpr int kNcclFloat32 = 7, kNcclSum = 0;               // ncclDataType_t / ncclRedOp_t values of rccl.h

}  // namespace

extern "C" int clora_comm_unique_id(void* id128) {
    if (!id128) return CLORA_ERR_ARG;
    if (!rccl_load()) return CLORA_ERR_LAUNCH;
    nccl_uid_t id;
    if (g_rccl.get_uid(&id) != 0) return CLORA_ERR_LAUNCH;
    memcpy(id128, id.internal, 128);
    return CLORA_OK;
}

extern "C" int clora_comm_init(const void* id128, int rank, int world) {
    if (!id128 || world < 1 || rank < 0 || rank >= world || g_rccl.comm) return CLORA_ERR_ARG;
    if (!rccl_load()) return CLORA_ERR_LAUNCH;
    nccl_uid_t id;
    memcpy(id.internal, id128, 128);
    nccl_comm_t c = nullptr;
    if (g_rccl.init_rank(&c, world, id, rank) != 0 || !c) return CLORA_ERR_LAUNCH;     // collective: every rank calls it
    g_rccl.comm = c; g_rccl.world = world; g_rccl.rank = rank;
    return CLORA_OK;
}

// which librccl the exchange is bound to (path of the object that defines the ncclAllReduce this library calls): a scaling record
// can then show that the C-ABI path and torch.distributed ran the same RCCL
extern "C" int clora_comm_library(char* path, size_t n) {
    if (!path || n == 0) return CLORA_ERR_ARG;
    path[0] = 0;
    if (!rccl_load()) return CLORA_ERR_LAUNCH;
    Dl_info di;
    if (!dladdr((void*)g_rccl.all_reduce, &di) || !di.dli_fname) return CLORA_ERR_LAUNCH;
    strncpy(path, di.dli_fname, n - 1);
    path[n - 1] = 0;
    return CLORA_OK;
}

extern "C" int clora_comm_world(void) { return g_rccl.comm ? g_rccl.world : 0; }
extern "C" int clora_comm_rank(void) { return g_rccl.comm ? g_rccl.rank : -1; }

extern "C" int clora_allreduce_flat_f32(float* buf, size_t n, void* stream) {
    if (!buf || n == 0) return CLORA_ERR_ARG;
    if (!g_rccl.comm) return CLORA_ERR_ARG;                                             // clora_comm_init first
    return g_rccl.all_reduce(buf, buf, n, kNcclFloat32, kNcclSum, g_rccl.comm, (hipStream_t)stream) == 0 ? CLORA_OK : CLORA_ERR_LAUNCH;
}

extern "C" int clora_comm_destroy(void) {
    if (!g_rccl.comm) return CLORA_OK;
    const int rc = g_rccl.destroy(g_rccl.comm);
    g_rccl.comm = nullptr; g_rccl.world = 0; g_rccl.rank = -1;
    return rc == 0 ? CLORA_OK : CLORA_ERR_LAUNCH;
}
