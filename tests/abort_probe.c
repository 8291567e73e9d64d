/* Test infrastructure (not product code): a SIGABRT probe for the GPU suite.  Round 6 saw ONE silent "Fatal Python error: Aborted"
 * in a full-suite run (two suites sharing the GPU): the main thread was in numpy code, another thread with no Python frame had
 * raised the signal, nothing was printed.  This handler runs ON THE THREAD that receives SIGABRT (abort() = raise() is
 * thread-directed), writes who sent it (si_code / si_pid) and that thread's C backtrace to a file, then hands over to the handler
 * that was installed before (pytest's faulthandler) so the Python stacks are still dumped.
 *   gcc -O1 -g -shared -fPIC tests/abort_probe.c -o /tmp/upamd_abort_probe.so ;  ctypes: upamd_abort_probe_install(b"/tmp/x.txt") */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static char g_path[512];
static struct sigaction g_old;

static void on_abort(int sig, siginfo_t *si, void *ctx) {
    int fd = open(g_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) {
        char line[256];
        int n = snprintf(line, sizeof line, "SIGABRT in pid %d tid %ld: si_code %d (%s) si_pid %d si_uid %d\n", (int)getpid(),
                         (long)syscall(SYS_gettid), si ? si->si_code : 0,
                         si && si->si_code == SI_TKILL ? "tkill: abort()/raise() inside this process" :
                         si && si->si_code == SI_USER ? "kill() from a process" : "other",
                         si ? (int)si->si_pid : -1, si ? (int)si->si_uid : -1);
        if (n > 0) (void)!write(fd, line, (size_t)n);
        void *bt[64];
        int depth = backtrace(bt, 64);
        backtrace_symbols_fd(bt, depth, fd);
        (void)!write(fd, "----\n", 5);
        close(fd);
    }
    sigaction(SIGABRT, &g_old, NULL);          /* previous handler (faulthandler) next, then the default action */
    raise(SIGABRT);
}

int upamd_abort_probe_install(const char *path) {
    struct sigaction sa;
    strncpy(g_path, path, sizeof g_path - 1);
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_abort;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    return sigaction(SIGABRT, &sa, &g_old);
}
