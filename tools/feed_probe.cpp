// tools/feed_probe.cpp -- a measurement, not part of the product: how fast do the bytes of a file in the page cache (tmpfs) reach HBM?
//   (a) pread by N threads into a pinned buffer, then hipMemcpyAsync (what BamFeed + agpu_ingest_push do today, without their overlap)
//   (b) mmap of the file, hipHostRegister of a piece, hipMemcpyAsync straight from the page cache, hipHostUnregister (no copy by the CPU)
//   (d) the same by T threads at once, each with its own pieces and its own stream: does pinning the pages of a mapping scale over threads?
// usage: feed_probe FILE [piece_MB] [threads] [modes, e.g. "abdc" (default) or "d": the pages of a mapping stay pinned once they were registered, so (d) says what it
//        should only in a process that has not run (b)]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(call) do { hipError_t e = (call); if (e != hipSuccess) { printf("%s: %s\n", #call, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
	if (argc < 2) return 2;
	const size_t piece = (size_t) (argc > 2 ? atoi(argv[2]) : 256) << 20;
	const int threads = argc > 3 ? atoi(argv[3]) : 16;
	const std::string modes = argc > 4 ? argv[4] : "abdc";
	const int only_workers = argc > 5 ? atoi(argv[5]) : 0;
	int fd = open(argv[1], O_RDONLY);
	struct stat st; fstat(fd, &st);
	const size_t size = (size_t) st.st_size / piece * piece;
	if (size == 0) return 3;
	void* device = nullptr; CHECK(hipMalloc(&device, piece));
	hipStream_t stream; CHECK(hipStreamCreate(&stream));
	void* pinned[2]; CHECK(hipHostMalloc(&pinned[0], piece)); CHECK(hipHostMalloc(&pinned[1], piece));
	// (a) pread + H2D, double-buffered
	for (int repeat = 0; repeat < 2 && modes.find('a') != std::string::npos; ++repeat) {
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
	for (int repeat = 0; repeat < 2 && modes.find('b') != std::string::npos; ++repeat) {
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
	// (d) T threads, each: register its piece, copy it on its own stream, unregister
	for (int workers : { 2, 4, 8, 16 }) {
		if (modes.find('d') == std::string::npos || (only_workers != 0 && workers != only_workers)) continue;
		std::vector<void*> targets(workers); std::vector<hipStream_t> streams(workers);
		for (int t = 0; t < workers; ++t) { CHECK(hipMalloc(&targets[t], piece)); CHECK(hipStreamCreate(&streams[t])); }
		const double t0 = now();
		std::vector<double> registering(workers, 0.0);
		std::vector<std::thread> pool;
		for (int t = 0; t < workers; ++t) pool.push_back(std::thread([&, t] {
			for (size_t at = piece * t; at < size; at += piece * workers) {
				const double r0 = now();
				if (hipHostRegister((char*) map + at, piece, hipHostRegisterDefault) != hipSuccess) return;
				registering[t] += now() - r0;
				if (hipMemcpyAsync(targets[t], (char*) map + at, piece, hipMemcpyHostToDevice, streams[t]) != hipSuccess) return;
				(void) hipStreamSynchronize(streams[t]);
				(void) hipHostUnregister((char*) map + at);
			}
		}));
		for (auto& t : pool) t.join();
		double longest = 0; for (double r : registering) if (r > longest) longest = r;
		printf("%d threads register + H2D + unregister their own pieces: %.2f GB in %.3f s = %.1f GB/s (the thread that spent most time registering: %.3f s)\n", workers, size / 1e9, now() - t0, size / 1e9 / (now() - t0), longest);
		for (int t = 0; t < workers; ++t) { (void) hipFree(targets[t]); (void) hipStreamDestroy(streams[t]); }
	}
	// (c) hipMemcpy from the mapping without registering it (the runtime stages it)
	if (modes.find('c') != std::string::npos) { const double t0 = now(); for (size_t at = 0; at < size; at += piece) CHECK(hipMemcpy(device, (char*) map + at, piece, hipMemcpyHostToDevice)); printf("hipMemcpy from the unregistered mapping: %.1f GB/s\n", size / 1e9 / (now() - t0)); }
	return 0;
}
