/* Diagnostic aid (not part of the product): a SIGSEGV / SIGBUS / SIGABRT handler that writes the faulting
 * address, the native back trace of the faulting thread (library + offset per frame) and the process' memory
 * map lines of executable segments to stderr, then hands over to whatever handler was installed before
 * (Python's faulthandler prints the Python stacks of all threads).  Build: gcc -shared -fPIC -O1 -g -o
 * libsegvbt.so segv_bt.c; load with ctypes.CDLL after faulthandler.enable() - tools/repro_trace_segv.sh. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static struct sigaction old_segv, old_bus, old_abrt;
static char altstack[1 << 16];

static void put(const char *s) { ssize_t r = write(2, s, strlen(s)); (void)r; }

static void handler(int sig, siginfo_t *si, void *uc)
{
    char line[256];
    void *frames[64];
    snprintf(line, sizeof line, "\n[segv_bt] signal %d, fault address %p, thread %ld\n", sig, si ? si->si_addr : 0,
             (long)syscall(SYS_gettid));
    put(line);
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    put("[segv_bt] executable mappings:\n");
    int fd = open("/proc/self/maps", O_RDONLY);
    if (fd >= 0) {
        static char buf[1 << 18];
        ssize_t got = 0, r;
        while (got < (ssize_t)sizeof buf - 1 && (r = read(fd, buf + got, sizeof buf - 1 - got)) > 0) got += r;
        buf[got] = 0;
        close(fd);
        for (char *p = buf; *p;) {
            char *e = strchr(p, '\n');
            if (!e) e = p + strlen(p);
            char keep = *e;
            *e = 0;
            if (strstr(p, " r-xp ") && strstr(p, ".so")) { put(p); put("\n"); }
            *e = keep;
            p = keep ? e + 1 : e;
        }
    }
    const struct sigaction *old = sig == SIGSEGV ? &old_segv : sig == SIGBUS ? &old_bus : &old_abrt;
    if ((old->sa_flags & SA_SIGINFO) && old->sa_sigaction) { old->sa_sigaction(sig, si, uc); }
    else if (old->sa_handler && old->sa_handler != SIG_DFL && old->sa_handler != SIG_IGN) { old->sa_handler(sig); }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    stack_t ss;
    ss.ss_sp = altstack;
    ss.ss_size = sizeof altstack;
    ss.ss_flags = 0;
    sigaltstack(&ss, 0);       /* (the calling thread only; other threads run the handler on their own stacks) */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, &sa, &old_bus);
    sigaction(SIGABRT, &sa, &old_abrt);
    void *warm[4];
    backtrace(warm, 4);        /* loads libgcc now, not inside the handler */
}
