// TEST INFRASTRUCTURE (oracle/): restatement of the reference's input-relation generator for string keys,
// create_strvec (strgen.cc:27-61), shared by oracle/ref_driver.cc (which feeds it to the compiled reference to
// make the goldens) and tests/cpp/test_dropin.cc (which feeds the same relations to the drop-in operator).
// Not part of the product.
//
// What the reference does, line by line (strgen.cc):
//   :31     sqrt_num = ceil(sqrt(number))
//   :33-44  the first sqrt_num lines of /usr/share/dict/words become {word_i, i}
//   :47-59  then, row-major over (i, j), {word_i + "-" + word_j, i + j} until `number` pairs exist
//   :51     std::random_shuffle(pairs)   (libstdc++: rand(); unseeded)
// Two departures, both forced: the word list is a parameter (the dictionary file does not exist in the build
// image or on the GPU box; tests/golden/words.txt is the fixture), and the shuffle is a Fisher-Yates walk
// driven by mix64(seed + i), so that it is reproducible everywhere (rand() is implementation-defined).
// As in the reference's benchmark (hashjoin_bench.cc:112-113) two calls give two relations over the SAME key
// set in different orders: every key matches exactly once.
#ifndef HMJ_STRGEN_RESTATED_H
#define HMJ_STRGEN_RESTATED_H
#include <cmath>
#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace hmj_strgen {

inline uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

inline std::vector<std::string> load_words(const std::string& path) {
  std::ifstream fs(path.c_str());
  if (!fs) throw std::runtime_error("word list not found: " + path);
  std::vector<std::string> w;
  std::string line;
  while (std::getline(fs, line)) w.push_back(line);
  return w;
}

inline std::vector<std::pair<std::string, uint64_t>> create_strvec(int number, const std::vector<std::string>& words,
                                                                   uint64_t seed) {
  std::vector<std::pair<std::string, uint64_t>> pairs;
  pairs.reserve(number);
  const int sqrt_num = (int)std::ceil(std::sqrt((double)number));
  if ((size_t)sqrt_num > words.size()) throw std::runtime_error("word list too short for this relation size");
  for (int i = 0; i < sqrt_num; i++) pairs.push_back(std::make_pair(words[i], (uint64_t)i));
  // (like the reference, a relation always holds the first sqrt_num plain words, even when number < sqrt_num^2;
  //  for number < sqrt_num -- sizes 0 and 1 -- the reference asserts out, here the list is cut)
  bool done = pairs.size() >= (size_t)number;
  for (int i = 0; i < sqrt_num && !done; i++)
    for (int j = 0; j < sqrt_num; j++) {
      if (pairs.size() == (size_t)number) {
        done = true;
        break;
      }
      pairs.push_back(std::make_pair(pairs[i].first + "-" + pairs[j].first, (uint64_t)(i + j)));
    }
  pairs.resize(number);
  for (size_t i = pairs.size(); i > 1; i--) {
    const size_t j = (size_t)(mix64(seed + (uint64_t)i) % (uint64_t)i);
    std::swap(pairs[i - 1], pairs[j]);
  }
  return pairs;
}

// FNV-1a over the relation as it stands: key bytes, then the 8 payload bytes (little endian), per element
inline uint64_t fnv_relation(const std::vector<std::pair<std::string, uint64_t>>& v) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (const auto& p : v) {
    for (unsigned char ch : p.first) {
      h ^= ch;
      h *= 0x100000001B3ull;
    }
    for (int b = 0; b < 8; b++) {
      h ^= (p.second >> (8 * b)) & 0xFF;
      h *= 0x100000001B3ull;
    }
  }
  return h;
}

}  // namespace hmj_strgen
#endif
