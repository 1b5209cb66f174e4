/* tools/abort_trace.c -- LD_PRELOAD shim: the call stack of the thread that raises SIGABRT (glibc's "double free or corruption" at the end of a test process says nothing else),
   printed with backtrace_symbols_fd before the default action.  gcc -shared -fPIC -o /tmp/libaborttrace.so tools/abort_trace.c */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_abort(int sig) {
	void* frames[64];
	int n = backtrace(frames, 64);
	static const char title[] = "ABORT TRACE (tools/abort_trace.c)\n";
	(void) !write(2, title, sizeof(title) - 1);
	backtrace_symbols_fd(frames, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}
/* (the first call of backtrace() loads libgcc_s and allocates: made here, not under the lock of the allocator that found the corruption) */
__attribute__((constructor)) static void install(void) { void* frames[4]; (void) backtrace(frames, 4); signal(SIGABRT, on_abort); }
