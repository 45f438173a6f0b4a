// rtc_check.cpp -- development: compiles one of the library's runtime-specialised kernels with hiprtc WITHOUT a GPU (the build
// container has none), exactly as rtc.cpp does (same sources, same options), and writes the code object for llvm-objdump.
//   rtc_check part|pscat <generated header file> <name expression> <out.co> [extra options...]
#include <hip/hiprtc.h>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
static std::string slurp(const std::string& p) { std::ifstream f(p); std::stringstream s; s << f.rdbuf(); return s.str(); }
int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: rtc_check part|pscat header expr out.co [opts]\n"); return 2; }
  const std::string kind = argv[1], dir = getenv("SSGPU_CSRC") ? getenv("SSGPU_CSRC") : "supersonic_amd/csrc";
  const std::string src = slurp(dir + (kind == "pscat" ? "/group_scatter_kernel.hip" : "/pipeline_kernels.hip"));
  const std::string vm = slurp(dir + "/vm.h"), launch = slurp(dir + "/launch.h"), body = slurp(dir + "/vm_body.inc"), gen = slurp(argv[2]);
  const char* headers[] = {vm.c_str(), launch.c_str(), body.c_str(), gen.c_str()};
  const char* names[] = {"vm.h", "launch.h", "vm_body.inc", kind == "pscat" ? "rtc_pscat.h" : "rtc_part.h"};
  hiprtcProgram p;
  if (hiprtcCreateProgram(&p, src.c_str(), "check.hip", 4, headers, names) != HIPRTC_SUCCESS) return 3;
  hiprtcAddNameExpression(p, argv[3]);
  std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics"};
  if (kind != "pscat") for (const char* o : {"-mllvm", "-structurizecfg-skip-uniform-regions", "-mllvm", "-amdgpu-use-divergent-register-indexing"}) opts.push_back(o);
  opts.push_back(kind == "pscat" ? "-DSSGPU_RTC_PSCAT" : "-DSSGPU_RTC_PART");
  for (int i = 5; i < argc; ++i) opts.push_back(argv[i]);
  const hiprtcResult r = hiprtcCompileProgram(p, (int)opts.size(), opts.data());
  size_t n = 0; hiprtcGetProgramLogSize(p, &n);
  if (n > 1) { std::string log(n, 0); hiprtcGetProgramLog(p, &log[0]); fprintf(stderr, "%s\n", log.c_str()); }
  if (r != HIPRTC_SUCCESS) { fprintf(stderr, "compilation failed (%d)\n", (int)r); return 1; }
  size_t cs = 0; hiprtcGetCodeSize(p, &cs); std::vector<char> code(cs); hiprtcGetCode(p, code.data());
  std::ofstream(argv[4], std::ios::binary).write(code.data(), (std::streamsize)cs);
  const char* low = nullptr; hiprtcGetLoweredName(p, argv[3], &low); printf("%s -> %s (%zu bytes)\n", argv[3], low ? low : "?", cs);
  return 0;
}
