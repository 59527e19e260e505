// What growing a device buffer costs: hipMalloc of N GB (today's workspace reservation) against reserving the address range once and
// mapping physical chunks into it as they are needed (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess).
// build: hipcc --offload-arch=gfx950 -O2 -o vmm_probe vmm_probe.hip ; run: ./vmm_probe [total_gb=16] [chunk_gb=1]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void touch(unsigned *p, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i;
}
__global__ void check(const unsigned *p, size_t n, unsigned v, unsigned long long *bad) {
  unsigned long long b = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += p[i] != v + (unsigned)i;
  if (b) atomicAdd(bad, b);
}

int main(int argc, char **argv) {
  const size_t GB = (size_t)1 << 30;
  const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 16) * GB, chunk_req = (size_t)(argc > 2 ? atoi(argv[2]) : 1) * GB;
  CHK(hipSetDevice(0));
  CHK(hipFree(nullptr));
  { // baseline: one hipMalloc of the whole size, first touch, free
    void *p = nullptr; double t0 = now();
    CHK(hipMalloc(&p, total)); double t1 = now();
    touch<<<4096, 256>>>((unsigned *)p, total / 4, 7u); CHK(hipDeviceSynchronize()); double t2 = now();
    CHK(hipFree(p)); double t3 = now();
    printf("hipMalloc %zu GB: %.1f ms (%.1f ms/GB), first touch %.1f ms, hipFree %.1f ms\n", total / GB, t1 - t0, (t1 - t0) / (total / (double)GB), t2 - t1, t3 - t2);
  }
  int vmm = 0;
  CHK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, 0));
  printf("virtual memory management supported: %d\n", vmm);
  if (!vmm) return 0;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0;
  CHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t chunk = (chunk_req + gran - 1) / gran * gran;
  printf("granularity %zu bytes, chunk %zu MB\n", gran, chunk >> 20);
  void *base = nullptr; double t0 = now();
  CHK(hipMemAddressReserve(&base, 4 * total, 0, nullptr, 0));
  printf("reserve %zu GB of addresses: %.2f ms\n", 4 * total / GB, now() - t0);
  hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> handles;
  double t_create = 0, t_map = 0, t_acc = 0, worst = 0;
  const size_t nchunks = total / chunk;
  unsigned long long *bad = nullptr; CHK(hipMalloc((void **)&bad, 8)); CHK(hipMemset(bad, 0, 8));
  for (size_t c = 0; c < nchunks; ++c) {
    hipMemGenericAllocationHandle_t h; double a = now();
    CHK(hipMemCreate(&h, chunk, &prop, 0)); double b = now();
    CHK(hipMemMap((char *)base + c * chunk, chunk, 0, h, 0)); double d = now();
    CHK(hipMemSetAccess((char *)base + c * chunk, chunk, &acc, 1)); double e = now();
    t_create += b - a; t_map += d - b; t_acc += e - d; worst = std::max(worst, e - a);
    handles.push_back(h);
    // the part mapped so far is one contiguous buffer: a kernel writes the newest chunk while the older ones keep their contents
    touch<<<2048, 256>>>((unsigned *)((char *)base + c * chunk), chunk / 4, (unsigned)c);
  }
  CHK(hipDeviceSynchronize());
  for (size_t c = 0; c < nchunks; ++c) check<<<2048, 256>>>((const unsigned *)((char *)base + c * chunk), chunk / 4, (unsigned)c, bad);
  unsigned long long hb = 1; CHK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
  printf("%zu chunks of %zu MB mapped one after the other: create %.1f + map %.1f + set access %.1f = %.1f ms in all (%.1f ms/GB), slowest chunk %.1f ms; wrong words after all maps: %llu\n",
         nchunks, chunk >> 20, t_create, t_map, t_acc, t_create + t_map + t_acc, (t_create + t_map + t_acc) / (total / (double)GB), worst, hb);
  // growth while a kernel runs on the part already mapped (does mapping wait for the device?)
  {
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int r = 0; r < 40; ++r) touch<<<4096, 256, 0, st>>>((unsigned *)base, total / 4, 1u);       // a few hundred ms of work
    hipMemGenericAllocationHandle_t h; double a = now();
    CHK(hipMemCreate(&h, chunk, &prop, 0));
    CHK(hipMemMap((char *)base + nchunks * chunk, chunk, 0, h, 0));
    CHK(hipMemSetAccess((char *)base + nchunks * chunk, chunk, &acc, 1)); double b = now();
    CHK(hipStreamSynchronize(st)); double c = now();
    printf("one more chunk mapped while kernels run on the mapped part: %.1f ms (the kernels finished %.1f ms later)\n", b - a, c - b);
    handles.push_back(h);
    CHK(hipStreamDestroy(st));
  }
  double u0 = now();
  CHK(hipMemUnmap(base, handles.size() * chunk));
  for (auto h : handles) CHK(hipMemRelease(h));
  CHK(hipMemAddressFree(base, 4 * total));
  printf("unmap + release + free addresses: %.1f ms\n", now() - u0);
  return 0;
}
