// idhmc_jit.hip -- user-supplied densities (IDHMC_MODEL_CUSTOM): the user's HIP source is compiled with hipRTC
// against the engine's own kernel templates (idhmc_general.hpp, idhmc_nuts_kernel.hpp), so a custom density
// runs through exactly the code paths of the built-in general density (dense MVN): evaluation, fused
// leapfrog, initial-stepsize search and the NUTS transition.  This is the device form of the reference's
// downward boundary, logdensity_and_gradient!(grad, model, q, sptr) (src/kinetic_energy.jl:73).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "idhmc_internal.hpp"

namespace idhmc {

int nuts_waves_per_block(int nch, int model, int shared_metric);
size_t nuts_lds_bytes(int L, bool lds_params, bool shared_metric, bool separable);

struct JitModule {
    hipModule_t mod = nullptr;
    hipFunction_t f_eval = nullptr, f_leapfrog = nullptr, f_stepsize = nullptr, f_nuts = nullptr, f_optimum = nullptr;
    size_t nuts_lds = 0;
};

static std::string library_dir()
{
    if (const char *e = getenv("IDHMC_SRC_DIR")) return e;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&library_dir), &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t k = p.find_last_of('/');
        return k == std::string::npos ? "." : p.substr(0, k);
    }
    return ".";
}

static void put_log(char *log, size_t cap, const std::string &s)
{
    if (!log || !cap) return;
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(log, s.data(), n);
    log[n] = 0;
}

int jit_build(const DevState &s, const char *source, JitModule **out, char *log, size_t log_cap)
{
    *out = nullptr;
    const std::string dir = library_dir();
    const bool shared = s.minv_stride == 0;
    std::string src = "#define IDHMC_JIT_USER_DENSITY 1\n#include \"idhmc_general.hpp\"\n#include \"idhmc_nuts_kernel.hpp\"\n#include \"idhmc_optimum.hpp\"\n"
                      "namespace idhmc {\n#line 1 \"user_density.hip\"\n";
    src += source;
    src += "\n}\n";
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "idhmc_custom_density.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        put_log(log, log_cap, "hiprtcCreateProgram failed");
        return 1;
    }
    const std::string n = std::to_string(s.nch);
    const std::string model = "idhmc::JitModel<" + n + ">";
    constexpr int kKernels = 5;
    const std::string names[kKernels] = {"idhmc::k_eval_general<" + n + ", " + model + ">",
                                  "idhmc::k_leapfrog_general<" + n + ", " + model + ">",
                                  "idhmc::k_stepsize_general<" + n + ", " + model + ">",
                                  "idhmc::k_nuts<" + n + ", " + model + ", " + (shared ? "true" : "false") + ">",
                                  "idhmc::k_local_optimum_general<" + n + ", " + model + ">"};
    for (const std::string &nm : names) hiprtcAddNameExpression(prog, nm.c_str());

    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::string arch = "gfx950";
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
        arch = prop.gcnArchName;
        const size_t colon = arch.find(':');
        if (colon != std::string::npos) arch = arch.substr(0, colon);
    }
    const std::string o_arch = "--offload-arch=" + arch;
    const std::string o_inc1 = "-I" + dir + "/csrc";
    const std::string o_inc2 = "-I" + dir + "/../include";
    std::vector<const char *> opts = {o_arch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", o_inc1.c_str(), o_inc2.c_str()};
    const hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string clog(ls, 0);
    if (ls) hiprtcGetProgramLog(prog, &clog[0]);
    if (rc != HIPRTC_SUCCESS) {
        put_log(log, log_cap, std::string("hipRTC: ") + hiprtcGetErrorString(rc) + "\n" + clog);
        hiprtcDestroyProgram(&prog);
        return 2;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    JitModule *m = new JitModule();
    if (hipModuleLoadData(&m->mod, code.data()) != hipSuccess) {
        put_log(log, log_cap, "hipModuleLoadData failed for the compiled density");
        hiprtcDestroyProgram(&prog);
        delete m;
        return 3;
    }
    hipFunction_t *fs[kKernels] = {&m->f_eval, &m->f_leapfrog, &m->f_stepsize, &m->f_nuts, &m->f_optimum};
    for (int i = 0; i < kKernels; ++i) {
        const char *lowered = nullptr;
        if (hiprtcGetLoweredName(prog, names[i].c_str(), &lowered) != HIPRTC_SUCCESS ||
            hipModuleGetFunction(fs[i], m->mod, lowered) != hipSuccess) {
            put_log(log, log_cap, "kernel " + names[i] + " not found in the compiled module");
            hiprtcDestroyProgram(&prog);
            (void)hipModuleUnload(m->mod);
            delete m;
            return 4;
        }
    }
    hiprtcDestroyProgram(&prog);
    m->nuts_lds = nuts_lds_bytes(s.L, false, shared, false);
    if (m->nuts_lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(m->f_nuts), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)m->nuts_lds) != hipSuccess) {
            (void)hipGetLastError();   // some runtimes accept large dynamic LDS for module kernels without the attribute
        }
    }
    *out = m;
    return 0;
}

void jit_destroy(JitModule *m)
{
    if (!m) return;
    if (m->mod) (void)hipModuleUnload(m->mod);
    delete m;
}

template <class Args>
static hipError_t launch_packed(hipFunction_t f, int grid, int block, size_t lds, hipStream_t st, Args &a)
{
    size_t sz = sizeof(Args);
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, (unsigned)lds, st, nullptr, extra);
}

static int general_grid_host(int64_t C)
{
    int64_t b = (C + 3) / 4;
    if (b > 256 * 8) b = 256 * 8;
    return (int)(b < 1 ? 1 : b);
}

hipError_t launch_eval_jit(const DevState &s, int random_q, hipStream_t st)
{
    const JitModule *m = static_cast<const JitModule *>(s.jit);
    if (!m) return hipErrorInvalidValue;
    struct { DevState s; int r; } a{s, random_q};
    return launch_packed(m->f_eval, general_grid_host(s.C), 256, 0, st, a);
}
hipError_t launch_leapfrog_jit(const DevState &s, double eps, int own, int n_steps, hipStream_t st)
{
    const JitModule *m = static_cast<const JitModule *>(s.jit);
    if (!m) return hipErrorInvalidValue;
    struct { DevState s; double eps; int own; int n; } a{s, eps, own, n_steps};
    return launch_packed(m->f_leapfrog, general_grid_host(s.C), 256, 0, st, a);
}
hipError_t launch_stepsize_search_jit(const DevState &s, hipStream_t st)
{
    const JitModule *m = static_cast<const JitModule *>(s.jit);
    if (!m) return hipErrorInvalidValue;
    struct { DevState s; } a{s};
    return launch_packed(m->f_stepsize, general_grid_host(s.C), 256, 0, st, a);
}
hipError_t launch_local_optimum_jit(const DevState &s, double penalty, int iterations, int grid, hipStream_t st)
{
    const JitModule *m = static_cast<const JitModule *>(s.jit);
    if (!m) return hipErrorInvalidValue;
    struct { DevState s; double penalty; int iterations; } a{s, penalty, iterations};
    return launch_packed(m->f_optimum, grid, 256, 0, st, a);
}
hipError_t launch_nuts_jit(const DevState &s, uint32_t iter, uint32_t flags, int grid, hipStream_t st)
{
    const JitModule *m = static_cast<const JitModule *>(s.jit);
    if (!m) return hipErrorInvalidValue;
    struct { DevState s; uint32_t iter; uint32_t flags; } a{s, iter, flags};
    return launch_packed(m->f_nuts, grid, nuts_waves_per_block(s.nch, s.model, s.minv_stride == 0) * 64, m->nuts_lds, st, a);
}

}  // namespace idhmc
