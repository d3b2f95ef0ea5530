/* Diagnostic preload: print the NATIVE backtrace of the thread that raises SIGABRT (Python's faulthandler only shows
 * Python frames, and chains to the handler that was installed before it -- this one).
 *   gcc -shared -fPIC -o /tmp/abort_trace.so tools/abort_trace.c && LD_PRELOAD=/tmp/abort_trace.so python -m pytest ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static int out_fd = 2; /* a copy of stderr taken at load time: pytest's fd capture redirects fd 2 itself */

static void on_abort(int sig) {
    void* bt[96];
    const char head[] = "\n=== abort_trace: native backtrace of the aborting thread ===\n";
    (void)!write(out_fd, head, sizeof(head) - 1);
    int n = backtrace(bt, 96);
    backtrace_symbols_fd(bt, n, out_fd);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    int d = dup(2);
    if (d >= 0) out_fd = d;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_abort;
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGABRT, &sa, 0);
    void* warm[2];
    backtrace(warm, 2); /* loads libgcc now, not inside the handler */
}
