/* LD_PRELOAD helper for crash hunts on the GPU box: native backtrace on SIGSEGV / SIGABRT / SIGBUS, then the default action.
   gcc -shared -fPIC -O1 -o tools/exp/segv_bt.so tools/exp/segv_bt.c */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t *info, void *ctx) {
  (void)ctx;
  static const char head[] = "\n==== native backtrace (segv_bt.so) ====\n";
  write(2, head, sizeof(head) - 1);
  void *frames[96];
  int n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, 2);
  (void)info;
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_fault;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGABRT, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
