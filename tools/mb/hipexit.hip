#include <hip/hip_runtime.h>
#include <unistd.h>
#include <stdio.h>
#include <chrono>
int main(int argc, char **argv) {
  size_t gb = argc > 1 ? atol(argv[1]) : 0;
  hipInit(0); hipSetDevice(0);
  void *p = nullptr;
  if (gb) { hipMalloc(&p, gb << 30); hipMemset(p, 0, gb << 30); hipDeviceSynchronize(); }
  double t = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
  printf("%.6f\n", t); fflush(stdout);
  _exit(0);
}
