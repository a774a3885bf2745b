// HipVector on TWO ranks over the host-staged transport (two processes sharing the GPU, a socket pair between them): ownership
// offsets, global indices, ghost refresh in close(), localize_to_all, and the reductions that PETSc performs over the communicator
// (VecMin / VecMax / VecNorm / VecDot; PetscVector.cpp, Parallel.hpp:351-377).  Global vector: x[i] = (i - 4.5)^2, N = 10;
// rank 0 owns 0..5 and ghosts global 7, rank 1 owns 6..9 and ghosts global 2 and 5.
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>
#include "HipBackend.hpp"

using namespace femus;
static int g_sock = -1, g_rank = 0;
static bool xfer(const double* send, int ns, double* recv, int nr) {     // rank 0 writes first, rank 1 reads first: no deadlock on big payloads
  auto wr = [&]() { size_t o = 0, n = (size_t)ns * 8; while (o < n) { ssize_t k = write(g_sock, (const char*)send + o, n - o); if (k <= 0) return false; o += k; } return true; };
  auto rd = [&]() { size_t o = 0, n = (size_t)nr * 8; while (o < n) { ssize_t k = read(g_sock, (char*)recv + o, n - o); if (k <= 0) return false; o += k; } return true; };
  return g_rank == 0 ? (wr() && rd()) : (rd() && wr());
}
static int exchange(void*, const double* send, const int* sc, double* recv, const int* rc) {
  const int o = 1 - g_rank;
  int so = 0, ro = 0;
  for (int r = 0; r < o; r++) { so += sc[r]; ro += rc[r]; }
  return xfer(send + so, sc[o], recv + ro, rc[o]) ? 0 : 1;
}
static int allreduce(void*, double* buf, int n) {
  std::vector<double> other(n);
  if (!xfer(buf, n, other.data(), n)) return 1;
  for (int k = 0; k < n; k++) buf[k] = g_rank == 0 ? buf[k] + other[k] : other[k] + buf[k];     // rank 0's value first on both ranks
  return 0;
}
#define CHECK(c) do { if (!(c)) { printf("rank %d: CHECK failed line %d: %s\n", g_rank, __LINE__, #c); fails++; } } while (0)

static int run() {
  int fails = 0;
  const int N = 10, first = g_rank == 0 ? 0 : 6, nloc = g_rank == 0 ? 6 : 4;
  const std::vector<int> ghosts = g_rank == 0 ? std::vector<int>{7} : std::vector<int>{2, 5};
  // exchange plan: rank 0 sends its entries 2 and 5 (local indices) to rank 1 and receives 1 value; rank 1 sends global 7 (local 1)
  const int sc0[2] = {0, 2}, rc0[2] = {0, 1}, si0[2] = {2, 5};
  const int sc1[2] = {1, 0}, rc1[2] = {2, 0}, si1[1] = {1};
  fh_halo_t halo = nullptr;
  hip_check(fh_halo_create_host(hip_context(), g_rank, 2, exchange, allreduce, nullptr, g_rank ? sc1 : sc0, g_rank ? si1 : si0, g_rank ? rc1 : rc0, &halo), "halo");
  HipVector x;
  x.init(N, nloc, ghosts, false, GHOSTED);
  x.attach_halo(halo);
  CHECK(x.first_local_index() == first && x.last_local_index() == first + nloc && x.size() == N && x.local_size() == nloc);
  std::vector<double> ref(N);
  for (int i = 0; i < N; i++) ref[i] = (i - 4.5) * (i - 4.5);
  for (int i = first; i < first + nloc; i++) x.set(i, ref[i]);         // GLOBAL indices on both ranks
  x.close();
  for (int i = first; i < first + nloc; i++) CHECK(x(i) == ref[i]);
  for (int g : ghosts) CHECK(x(g) == ref[g]);                           // ghosts refreshed from their owners
  std::vector<double> all;
  x.localize_to_all(all);
  CHECK((int)all.size() == N);
  for (int i = 0; i < N; i++) CHECK(all[i] == ref[i]);
  double s = 0, s2 = 0, mx = -1e300, mn = 1e300;
  for (double v : ref) { s += v; s2 += v * v; mx = std::max(mx, v); mn = std::min(mn, v); }
  CHECK(x.max() == mx && x.min() == mn && x.linfty_norm() == mx);
  CHECK(std::fabs(x.sum() - s) < 1e-13 * s && std::fabs(x.l1_norm() - s) < 1e-13 * s && std::fabs(x.l2_norm() - std::sqrt(s2)) < 1e-13 * std::sqrt(s2));
  HipVector y;
  y.init(x);
  CHECK(y.first_local_index() == first && y.halo() == halo);
  y = static_cast<const NumericVector&>(x);
  CHECK(std::fabs(y.dot(x) - s2) < 1e-13 * s2);
  // staged adds with global indices: owned entries and a ghost entry, applied by close(); the add to the ghost entry is shipped to its owner
  // and summed there (VecSetValues ADD_VALUES + VecAssemblyBegin/End), then the ghost copy is refreshed (VecGhostUpdate INSERT_VALUES, SCATTER_FORWARD)
  const int o0 = first, gh = ghosts[0];
  y.add_vector_blocked(std::vector<double>{1., 2., 100.}, std::vector<int>{o0, o0, gh});
  y.add(o0 + 1, 0.5);
  y.close();
  CHECK(y(o0) == ref[o0] + 3. && y(o0 + 1) == ref[o0 + 1] + 0.5 + (g_rank == 1 ? 100. : 0.));     // rank 1's o0 + 1 = global 7 also received rank 0's 100
  const double expect_ghost = ref[gh] + 100. + (gh == 7 ? 0.5 : 0.);      // global 7 is rank 1's o0 + 1 (its own + 0.5); global 2 got rank 1's 100 only
  CHECK(y(gh) == expect_ghost);
  std::vector<double> one;
  y.localize_to_one(one, 0);
  CHECK((int)one.size() == N && one[0] == ref[0] + 3. && one[6] == ref[6] + 3. && one[7] == ref[7] + 100.5 && one[2] == ref[2] + 100. && one[1] == ref[1] + 0.5);
  x.clear();
  y.clear();
  fh_halo_destroy(halo);
  printf("rank %d: %s\n", g_rank, fails ? "FAILED" : "ok");
  return fails;
}

int main() {
  int sv[2];
  if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) return 2;
  const pid_t pid = fork();          // before any HIP call: each process creates its own context on the shared device
  if (pid < 0) return 2;
  g_rank = pid == 0 ? 1 : 0;
  g_sock = sv[g_rank];
  close(sv[1 - g_rank]);
  const int fails = run();
  fflush(stdout);
  if (pid == 0) _exit(fails ? 1 : 0);
  int st = 0;
  waitpid(pid, &st, 0);
  const bool ok = !fails && WIFEXITED(st) && WEXITSTATUS(st) == 0;
  std::cout << (ok ? "TWO RANKS OK" : "TWO RANKS FAILED") << std::endl;
  return ok ? 0 : 1;
}
