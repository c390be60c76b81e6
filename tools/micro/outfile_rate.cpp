// How fast can one output file take bytes?  pwrite from T threads vs. a shared mapping filled by T threads (tmpfs or a disk file).
// usage: outfile_rate <path> <GB> <threads>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
	if (argc < 4) return 1;
	const size_t n = (size_t)(atof(argv[2]) * (1 << 30)); const int T = atoi(argv[3]);
	const size_t P = 32u << 20;                                  // piece, as the stream driver's ring
	char *src = (char *)malloc(P); memset(src, 'A', P);
	for (int mode = 0; mode < 3; ++mode) {
		unlink(argv[1]);
		const int fd = open(argv[1], O_CREAT | O_RDWR | O_TRUNC, 0644); if (fd < 0) return 2;
		const double t0 = now();
		if (mode == 0) {
			for (size_t off = 0; off < n; off += P) {
				std::vector<std::thread> th; const size_t per = P / T;
				for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t lo = t * per, hi = t == T - 1 ? P : lo + per; while (lo < hi) { ssize_t w = pwrite(fd, src + lo, hi - lo, off + lo); if (w <= 0) break; lo += w; } });
				for (auto &x : th) x.join();
			}
		} else {
			const size_t W = mode == 1 ? ((size_t)1 << 30) : P;        // mapping window: 1 GB at a time, or piece by piece
			for (size_t wo = 0; wo < n; wo += W) {
				const size_t wl = wo + W <= n ? W : n - wo;
				if (ftruncate(fd, wo + wl)) return 3;
				char *m = (char *)mmap(nullptr, wl, PROT_READ | PROT_WRITE, MAP_SHARED, fd, wo); if (m == MAP_FAILED) return 4;
				for (size_t off = 0; off < wl; off += P) {
					const size_t pl = off + P <= wl ? P : wl - off;
					std::vector<std::thread> th; const size_t per = (pl / T + 4095) & ~(size_t)4095;
					for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t lo = t * per, hi = lo + per < pl ? lo + per : pl; if (lo < hi) memcpy(m + off + lo, src + lo, hi - lo); });
					for (auto &x : th) x.join();
				}
				munmap(m, wl);
			}
		}
		const double t1 = now(); close(fd);
		printf("%s: %.2f GB in %.3f s = %.2f GB/s (%d threads)\n", mode == 0 ? "pwrite " : mode == 1 ? "mmap 1G" : "mmap 32M", n / 1e9, t1 - t0, n / 1e9 / (t1 - t0), T);
	}
	unlink(argv[1]);
	return 0;
}
