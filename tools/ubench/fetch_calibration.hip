// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library.
// MI355X_MICROARCH.md (HBM): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read (16 B per lane);
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".
// Every kernel below moves a KNOWN number of bytes, once, over a buffer four times the Infinity Cache:
//   read16 / write16      16 B per lane, whole wavefront contiguous (the tile kernels' d4 operand / hand-off accesses)
//   read8  / write8        8 B per lane, whole wavefront contiguous
//   rows16_read / _write   one 16-lane DPP row per "sequence": 128 B contiguous per row and instruction, the four rows
//                          of a wavefront in four different sequences (the n <= 15 kernels' full-row accesses)
//   rows10_read / _write   the same with 10 active lanes: 80 B contiguous, unaligned (n = 10 matrices row by row)
// Build + run:  hipcc -O2 --offload-arch=gfx950 tools/ubench/fetch_calibration.hip -o variants/fetch_cal
//               rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -o cal -- variants/fetch_cal
//               rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/write -o cal -- variants/fetch_cal
//               python tools/ubench/fetch_calibration.py out          (expected bytes: stdout of the binary)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void cal_read16(const d2* __restrict__ p, long n2, double* sink) {
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
    const d2 v = p[i];
    acc += v[0] + v[1];
  }
  if (acc == 12345.678) sink[0] = acc;
}

__global__ void cal_read8(const double* __restrict__ p, long n, double* sink) {
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678) sink[0] = acc;
}

__global__ void cal_write16(d2* __restrict__ p, long n2) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x)
    p[i] = d2{1.0, 2.0};
}

__global__ void cal_write8(double* __restrict__ p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 3.0;
}

// one 16-lane row per sequence; LANES active lanes read/write LANES contiguous doubles per instruction
template <int LANES, bool WRITE>
__global__ void cal_rows(double* __restrict__ p, int seqs, long seq_doubles, double* sink) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, r16 = threadIdx.x & 15;
  if (row >= seqs) return;
  double* s = p + (long)row * seq_doubles;
  double acc = 0.0;
  const long chunks = seq_doubles / LANES;
#pragma unroll 4
  for (long c = 0; c < chunks; ++c) {
    if (r16 < LANES) {
      if (WRITE) s[c * LANES + r16] = 4.0;
      else acc += s[c * LANES + r16];
    }
  }
  if (!WRITE && acc == 12345.678) sink[0] = acc;
}

int main() {
  const long N = 1L << 27;                       // doubles: 1 GiB
  const int SEQS = 4096;
  const long SEQ16 = N / SEQS;                   // 32768 doubles per sequence
  const long SEQ10 = (SEQ16 / 10) * 10;          // 32760
  double *buf, *sink;
  CK(hipMalloc(&buf, N * sizeof(double)));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 0, N * sizeof(double)));
  CK(hipDeviceSynchronize());
  const int G = 256 * 16;
  for (int rep = 0; rep < 3; ++rep) {
    cal_read16<<<G, 256>>>((const d2*)buf, N / 2, sink);
    cal_read8<<<G, 256>>>(buf, N, sink);
    cal_rows<16, false><<<SEQS * 16 / 256, 256>>>(buf, SEQS, SEQ16, sink);
    cal_rows<10, false><<<SEQS * 16 / 256, 256>>>(buf, SEQS, SEQ10, sink);
    cal_write16<<<G, 256>>>((d2*)buf, N / 2);
    cal_write8<<<G, 256>>>(buf, N);
    cal_rows<16, true><<<SEQS * 16 / 256, 256>>>(buf, SEQS, SEQ16, sink);
    cal_rows<10, true><<<SEQS * 16 / 256, 256>>>(buf, SEQS, SEQ10, sink);
    CK(hipDeviceSynchronize());
  }
  printf("expected_bytes cal_read16 %ld\n", N * 8);
  printf("expected_bytes cal_read8 %ld\n", N * 8);
  printf("expected_bytes cal_rows<16,false> %ld\n", (long)SEQS * SEQ16 * 8);
  printf("expected_bytes cal_rows<10,false> %ld\n", (long)SEQS * SEQ10 * 8);
  printf("expected_bytes cal_write16 %ld\n", N * 8);
  printf("expected_bytes cal_write8 %ld\n", N * 8);
  printf("expected_bytes cal_rows<16,true> %ld\n", (long)SEQS * SEQ16 * 8);
  printf("expected_bytes cal_rows<10,true> %ld\n", (long)SEQS * SEQ10 * 8);
  return 0;
}
