#define _GNU_SOURCE
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#include <stdint.h>
static uint64_t* samples; static volatile long n_samples; enum { CAP = 1 << 22 };
static void handler(int sig, siginfo_t* si, void* uc_) { ucontext_t* uc = (ucontext_t*)uc_; long i = n_samples; if (i < CAP) { samples[i] = (uint64_t)uc->uc_mcontext.gregs[REG_RIP]; n_samples = i + 1; } }
void sprof_start(void) { if (!samples) samples = (uint64_t*)malloc(sizeof(uint64_t) * CAP); struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = handler; sa.sa_flags = SA_SIGINFO | SA_RESTART; sigaction(SIGPROF, &sa, 0);
  struct itimerval it; it.it_interval.tv_sec = 0; it.it_interval.tv_usec = 500; it.it_value = it.it_interval; setitimer(ITIMER_PROF, &it, 0); }
void sprof_stop(const char* path) { struct itimerval it; memset(&it, 0, sizeof it); setitimer(ITIMER_PROF, &it, 0); FILE* f = fopen(path, "w"); FILE* m = fopen("/proc/self/maps", "r"); char line[512];
  while (fgets(line, sizeof line, m)) if (strstr(line, " r-xp ")) fprintf(f, "M %s", line); fclose(m); for (long i = 0; i < n_samples; i++) fprintf(f, "S %lx\n", samples[i]); fclose(f); n_samples = 0; }
