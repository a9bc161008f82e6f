// ref_pin.cpp — links the ONE reference file on the hot path that compiles
// from its own sources in this image: utility/HashPair.hpp (std headers only).
// Built by oracle/Makefile into oracle/_ref/ref_pin with -I/root/reference; the
// reference source is included where it lies, never copied.  Usage:
//   ref_pin a b [a b ...]   -> prints quickstep::CombineHashes(a, b) per pair (hex)
// tests/test_oracle_pins.py checks oracle/qsx_oracle.cpp:combine_hashes against it.
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "utility/HashPair.hpp"

int main(int argc, char **argv) {
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::size_t a = std::strtoull(argv[i], nullptr, 0);
    const std::size_t b = std::strtoull(argv[i + 1], nullptr, 0);
    std::printf("%016llx\n", static_cast<unsigned long long>(quickstep::CombineHashes(a, b)));
  }
  return 0;
}
