// Unity build of the library's three translation units (rl_host.h): used by experiment builds that
// need ONE code object -- the RL_TIMING phase stamps live in one __device__ buffer -- and handy for
// a one-command build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o librunlmc_hip.so runlmc_hip.hip
// The product build (python -m runlmc_amd.build) compiles the three files separately, in parallel.
#include "rl_gridop.hip"
#include "rl_ski.hip"
#include "rl_solve.hip"
