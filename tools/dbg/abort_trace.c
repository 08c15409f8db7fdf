// Debug aid: prints a C backtrace to stderr when the process receives SIGABRT (a HIP runtime abort() inside a library call
// shows only Python frames in faulthandler).  gcc -shared -fPIC -o abort_trace.so abort_trace.c; load with ctypes.CDLL.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_abort(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) { signal(SIGABRT, on_abort); }
