// tests/cpp/sanitize_host_test.cpp — the pure-host parts of the library (the DSPBuffer ring of dspbuffer.cpp, the
// coefficient makers of coeffs.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer, and the ring's single-producer /
// single-consumer contract under ThreadSanitizer (tests/test_host_cpp.py builds it three ways with g++ and runs it; no
// GPU involved). Random operation sequences against a trivially correct model: a std::deque of floats.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <random>
#include <thread>
#include <vector>

#include "mlgpu.h"

static int failures = 0;
#define REQUIRE(cond)                                                 \
  do                                                                  \
  {                                                                   \
    if (!(cond))                                                      \
    {                                                                 \
      printf("REQUIRE failed at line %d: %s\n", __LINE__, #cond);     \
      ++failures;                                                     \
    }                                                                 \
  } while (0)

static void ringAgainstModel(unsigned seed)
{
  std::mt19937 rng(seed);
  mlgpu_dspbuffer* b = mlgpu_dspbuffer_create();
  const int want = 64 + (int)(rng() % 3000);
  const size_t size = mlgpu_dspbuffer_resize(b, want);
  REQUIRE(size >= (size_t)want && (size & (size - 1)) == 0);
  REQUIRE(mlgpu_dspbuffer_resize(b, -5) == 0);                 // refused, not a shift by a negative count
  REQUIRE(mlgpu_dspbuffer_resize(b, (1 << 30) + 1) == 0);      // 1 << 31 would overflow the int the size is computed in
  REQUIRE(mlgpu_dspbuffer_resize(b, want) == size);
  std::deque<float> model;
  float next = 1.f;
  std::vector<float> tmp(2 * size + 64), got(2 * size + 64);
  for (int step = 0; step < 4000; ++step)
  {
    const unsigned op = rng() % 6;
    if (op <= 1)
    {
      const size_t n = rng() % (size + size / 2);                // sometimes more than fits: the oldest data is clobbered
      for (size_t i = 0; i < n; ++i) tmp[i] = next++;
      mlgpu_dspbuffer_write(b, tmp.data(), n);
      for (size_t i = 0; i < n; ++i) model.push_back(tmp[i]);
      // the reference keeps the newest `size` samples at most (MLDSPBuffer.h:162-167); a write larger than the ring keeps its tail
      while (model.size() > size) model.pop_front();
      if (n > size) { mlgpu_dspbuffer_clear(b); model.clear(); }  // undefined region of the reference: start again
    }
    else if (op == 2)
    {
      const size_t n = rng() % (size + 8);
      const size_t r = mlgpu_dspbuffer_read(b, got.data(), n);
      REQUIRE(r == (n < model.size() ? n : model.size()));
      for (size_t i = 0; i < r; ++i)
      {
        REQUIRE(got[i] == model.front());
        model.pop_front();
      }
    }
    else if (op == 3)
    {
      float v[64];
      const int ok = mlgpu_dspbuffer_read_vector(b, v);
      REQUIRE((ok != 0) == (model.size() >= 64));
      if (ok)
        for (int i = 0; i < 64; ++i)
        {
          REQUIRE(v[i] == model.front());
          model.pop_front();
        }
    }
    else if (op == 4)
    {
      const size_t n = rng() % 200;
      mlgpu_dspbuffer_discard(b, n);
      for (size_t i = 0; i < n && !model.empty(); ++i) model.pop_front();
    }
    else
    {
      const size_t n = model.empty() ? 0 : 1 + rng() % model.size();
      mlgpu_dspbuffer_peek_most_recent(b, got.data(), n);
      for (size_t i = 0; i < n; ++i) REQUIRE(got[i] == model[model.size() - n + i]);
    }
    REQUIRE(mlgpu_dspbuffer_read_available(b) == model.size());
  }
  mlgpu_dspbuffer_destroy(b);
}

// one writer thread, one reader thread, no lock: every float arrives once, in order (run under -fsanitize=thread too)
static void ringTwoThreads()
{
  mlgpu_dspbuffer* b = mlgpu_dspbuffer_create();
  mlgpu_dspbuffer_resize(b, 1024);
  const size_t total = 400000;
  std::atomic<bool> done{false};
  std::thread writer([&] {
    size_t sent = 0;
    float chunk[97];
    while (sent < total)
    {
      const size_t n = (total - sent < 97) ? total - sent : 97;
      if (mlgpu_dspbuffer_write_available(b) < n) continue;      // a real-time producer never overruns its consumer
      for (size_t i = 0; i < n; ++i) chunk[i] = (float)((sent + i) & 0xFFFFF);
      mlgpu_dspbuffer_write(b, chunk, n);
      sent += n;
    }
    done = true;
  });
  size_t seen = 0;
  float buf[128];
  while (seen < total)
  {
    const size_t r = mlgpu_dspbuffer_read(b, buf, 128);
    for (size_t i = 0; i < r; ++i) REQUIRE(buf[i] == (float)((seen + i) & 0xFFFFF));
    seen += r;
  }
  writer.join();
  REQUIRE(done.load() && mlgpu_dspbuffer_read_available(b) == 0);
  mlgpu_dspbuffer_destroy(b);
}

static void coefficientMakers()
{
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(0.f, 1.f);
  float c[8];
  for (int i = 0; i < 20000; ++i)
  {
    const float omega = 0.0001f + 0.49f * u(rng), k = 0.01f + 3.f * u(rng), A = 0.1f + 8.f * u(rng);
    mlgpu_lopass_make_coeffs(omega, k, c);
    mlgpu_hipass_make_coeffs(omega, k, c);
    mlgpu_bandpass_make_coeffs(omega, k, c);
    mlgpu_loshelf_make_coeffs(omega, k, A, c);
    mlgpu_hishelf_make_coeffs(omega, k, A, c);
    mlgpu_bell_make_coeffs(omega, k, A, c);
    mlgpu_onepole_make_coeffs(omega, c);
    mlgpu_adsr_calc_coeffs(0.001f + u(rng), 0.001f + u(rng), u(rng), 0.001f + u(rng), 48000.f, c);
    REQUIRE(c[0] == c[0]);
  }
}

int main(int argc, char** argv)
{
  const bool threadsOnly = argc > 1 && !strcmp(argv[1], "threads");
  if (!threadsOnly)
  {
    for (unsigned s = 1; s <= 12; ++s) ringAgainstModel(s);
    coefficientMakers();
  }
  ringTwoThreads();
  if (failures == 0) printf("All tests passed\n");
  return failures ? 1 : 0;
}
