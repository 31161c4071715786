/* LD_PRELOAD helper: prints a native backtrace on SIGSEGV (python's faulthandler shows Python frames only).
 * build: gcc -shared -fPIC -O1 -o tools/bin/segv_bt.so tools/segv_bt.c
 * use:   LD_PRELOAD=$PWD/tools/bin/segv_bt.so python bench.py ; resolve the offsets with llvm-symbolizer --obj=<library> */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <string.h>
static void handler(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    const char msg[] = "=== SIGSEGV backtrace ===\n";
    write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void init(void) {
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler; sa.sa_flags = SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
}
