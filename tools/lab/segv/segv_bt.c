// LD_PRELOAD helper for the GPU box (no gdb there): a SIGSEGV / SIGBUS / SIGABRT prints the faulting address, the native backtrace of the faulting thread and where
// the product's library is mapped, then dies of the signal as before.   gcc -O1 -g -shared -fPIC -o libsegv_bt.so segv_bt.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
static char alt[1 << 16];
static void on_fault(int sig, siginfo_t *si, void *uc) {
    (void)uc;
    char line[256];
    int n = snprintf(line, sizeof line, "\n[segv_bt] signal %d, address %p, thread %ld\n", sig, si ? si->si_addr : 0, (long)gettid());
    write(2, line, (size_t)n);
    void *bt[64];
    int k = backtrace(bt, 64);
    backtrace_symbols_fd(bt, k, 2);
    int fd = open("/proc/self/maps", O_RDONLY);
    if (fd >= 0) {
        static char buf[1 << 20];
        ssize_t got = 0, r;
        while ((r = read(fd, buf + got, sizeof buf - 1 - (size_t)got)) > 0) got += r;
        buf[got] = 0;
        for (char *p = buf; *p;) {
            char *e = strchr(p, '\n'); if (!e) break; *e = 0;
            if (strstr(p, "regtools") && strstr(p, "r-xp")) { write(2, "[segv_bt] ", 10); write(2, p, strlen(p)); write(2, "\n", 1); }
            p = e + 1;
        }
        close(fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault; sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_RESETHAND;
    sigaction(SIGSEGV, &sa, 0); sigaction(SIGBUS, &sa, 0);
}
