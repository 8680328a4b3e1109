/* LD_PRELOAD helper for the GPU box (no gdb there): prints the native backtrace of the aborting thread on SIGABRT /
 * SIGSEGV, then re-raises.  gcc -shared -fPIC -o abrt_trace.so abrt_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <stdlib.h>
static void handler(int sig) {
    void* bt[64];
    const char* path = getenv("ABRT_TRACE_FILE");
    int fd = path ? open(path, O_WRONLY | O_CREAT | O_APPEND, 0644) : 2;
    if (fd < 0) fd = 2;
    const char msg[] = "\n==== native backtrace (abrt_trace.so) ====\n";
    if (write(fd, msg, sizeof msg - 1) < 0) { }
    int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, fd);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void init(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sigaction(SIGABRT, &sa, NULL);
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
}
