// tools/feed_probe.cpp -- a measurement, not part of the product: how fast do the bytes of a file in the page cache (tmpfs) reach HBM?
//   (a) pread by N threads into a pinned buffer, then hipMemcpyAsync (what BamFeed + agpu_ingest_push do today, without their overlap)
//   (b) mmap of the file, hipHostRegister of a piece, hipMemcpyAsync straight from the page cache, hipHostUnregister (no copy by the CPU)
// usage: feed_probe FILE [piece_MB] [threads]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(call) do { hipError_t e = (call); if (e != hipSuccess) { printf("%s: %s\n", #call, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
	if (argc < 2) return 2;
	const size_t piece = (size_t) (argc > 2 ? atoi(argv[2]) : 256) << 20;
	const int threads = argc > 3 ? atoi(argv[3]) : 16;
	int fd = open(argv[1], O_RDONLY);
	struct stat st; fstat(fd, &st);
	const size_t size = (size_t) st.st_size / piece * piece;
	if (size == 0) return 3;
	void* device = nullptr; CHECK(hipMalloc(&device, piece));
	hipStream_t stream; CHECK(hipStreamCreate(&stream));
	void* pinned[2]; CHECK(hipHostMalloc(&pinned[0], piece)); CHECK(hipHostMalloc(&pinned[1], piece));
	// (a) pread + H2D, double-buffered
	for (int repeat = 0; repeat < 2; ++repeat) {
		const double t0 = now(); double read_seconds = 0;
		for (size_t at = 0, k = 0; at < size; at += piece, ++k) {
			const double r0 = now();
			std::vector<std::thread> pool;
			for (int t = 0; t < threads; ++t) pool.push_back(std::thread([&, t] { const size_t from = piece * t / threads, to = piece * (t + 1) / threads; size_t done = 0; while (done < to - from) { ssize_t n = pread(fd, (char*) pinned[k & 1] + from + done, to - from - done, at + from + done); if (n <= 0) break; done += n; } }));
			for (auto& t : pool) t.join();
			read_seconds += now() - r0;
			CHECK(hipStreamSynchronize(stream)); // (the copy of the piece before, from the other buffer)
			CHECK(hipMemcpyAsync(device, pinned[k & 1], piece, hipMemcpyHostToDevice, stream));
		}
		CHECK(hipStreamSynchronize(stream));
		printf("pread x%d + H2D: %.2f GB in %.3f s = %.1f GB/s (the reads alone %.3f s = %.1f GB/s)\n", threads, size / 1e9, now() - t0, size / 1e9 / (now() - t0), read_seconds, size / 1e9 / read_seconds);
	}
	// (b) mmap + register + H2D + unregister
	void* map = mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
	if (map == MAP_FAILED) { printf("mmap failed\n"); return 1; }
	for (int repeat = 0; repeat < 2; ++repeat) {
		const double t0 = now(); double register_seconds = 0, copy_seconds = 0, unregister_seconds = 0;
		for (size_t at = 0; at < size; at += piece) {
			const double r0 = now();
			hipError_t e = hipHostRegister((char*) map + at, piece, hipHostRegisterDefault);
			if (e != hipSuccess) { printf("hipHostRegister of a mapped file: %s\n", hipGetErrorString(e)); return 0; }
			const double r1 = now();
			CHECK(hipMemcpyAsync(device, (char*) map + at, piece, hipMemcpyHostToDevice, stream));
			CHECK(hipStreamSynchronize(stream));
			const double r2 = now();
			CHECK(hipHostUnregister((char*) map + at));
			register_seconds += r1 - r0; copy_seconds += r2 - r1; unregister_seconds += now() - r2;
		}
		printf("mmap + register + H2D + unregister: %.2f GB in %.3f s = %.1f GB/s (register %.3f s, copy %.3f s = %.1f GB/s, unregister %.3f s)\n", size / 1e9, now() - t0, size / 1e9 / (now() - t0), register_seconds, copy_seconds, size / 1e9 / copy_seconds, unregister_seconds);
	}
	// (c) hipMemcpy from the mapping without registering it (the runtime stages it)
	{ const double t0 = now(); for (size_t at = 0; at < size; at += piece) CHECK(hipMemcpy(device, (char*) map + at, piece, hipMemcpyHostToDevice)); printf("hipMemcpy from the unregistered mapping: %.1f GB/s\n", size / 1e9 / (now() - t0)); }
	return 0;
}
