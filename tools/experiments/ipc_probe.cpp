// ipc_probe.cpp -- do hipIpc* between two processes on ONE GPU and the stream memory operations work on this stack?
// (feasibility probe for a stream-ordered multi-process stand-in of RCCL; tools only)
// g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/experiments/ipc_probe.cpp -o build/ipc_probe -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("[%d] %s failed: %s (line %d)\n", getpid(), #x, hipGetErrorString(e), __LINE__); fflush(stdout); _exit(2); } } while (0)

struct Shared { std::atomic<int> stage; hipIpcMemHandle_t handle; unsigned flag_parent, flag_child; };

int main()
{
    const char *name = "/bq_ipc_probe";
    shm_unlink(name);
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, 4096) != 0) { printf("shm failed\n"); return 1; }
    Shared *sh = (Shared *)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    memset(sh, 0, sizeof *sh);
    const size_t n = 1 << 20;
    pid_t pid = fork();                                  // before any HIP call
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(hipHostRegister(sh, 4096, hipHostRegisterMapped));
    unsigned *dflag_parent = nullptr, *dflag_child = nullptr;
    CK(hipHostGetDevicePointer((void **)&dflag_parent, &sh->flag_parent, 0));
    CK(hipHostGetDevicePointer((void **)&dflag_child, &sh->flag_child, 0));
    if (pid != 0) {                                      // parent: owns the mailbox
        float *box; CK(hipMalloc(&box, n * 4)); CK(hipMemset(box, 0, n * 4));
        CK(hipIpcGetMemHandle(&sh->handle, box));
        sh->stage.store(1);
        // stream-ordered: wait until the child says the mailbox is full, then read it back
        CK(hipStreamWaitValue32(st, dflag_child, 1, hipStreamWaitValueGte, 0xffffffffu));
        std::vector<float> host(n);
        CK(hipMemcpyAsync(host.data(), box, n * 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamWriteValue32(st, dflag_parent, 1, 0));
        CK(hipStreamSynchronize(st));
        int status = 0; waitpid(pid, &status, 0);
        bool ok = true;
        for (size_t i = 0; i < n; i += 4097) ok = ok && host[i] == (float)(i % 1000);
        printf("parent: data %s, child exit %d\n", ok ? "OK" : "WRONG", WEXITSTATUS(status));
        shm_unlink(name);
        return ok && WEXITSTATUS(status) == 0 ? 0 : 1;
    }
    while (sh->stage.load() < 1) usleep(100);
    float *peer = nullptr;
    CK(hipIpcOpenMemHandle((void **)&peer, sh->handle, hipIpcMemLazyEnablePeerAccess));
    float *src; CK(hipMalloc(&src, n * 4));
    std::vector<float> host(n);
    for (size_t i = 0; i < n; i++) host[i] = (float)(i % 1000);
    CK(hipMemcpy(src, host.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpyAsync(peer, src, n * 4, hipMemcpyDeviceToDevice, st));
    CK(hipStreamWriteValue32(st, dflag_child, 1, 0));
    CK(hipStreamWaitValue32(st, dflag_parent, 1, hipStreamWaitValueGte, 0xffffffffu));
    CK(hipStreamSynchronize(st));
    CK(hipIpcCloseMemHandle(peer));
    printf("child: done\n");
    return 0;
}
