// Sanitizer fuzzing of the RDKit-free ingestion (host code only, no GPU): random byte edits of real SMILES / SD files are
// fed to nvmk_smiles_parse_text / nvmk_sdf_parse_text from exactly-sized heap buffers (so that AddressSanitizer sees any read
// past the end), both with and without NVMK_SMILES_PERCEIVE_AROMATICITY, and every accessor is called on the result.
//   ./run.sh            builds with -fsanitize=address,undefined and runs over tests/golden
//   fuzz_ingestion smi|sdf <file> <rounds>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/nvmolkit_amd.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const bool    sdf = std::strcmp(argv[1], "sdf") == 0;
  std::ifstream f(argv[2], std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string base = ss.str();
  std::mt19937      rng(12345);
  const char        alphabet[] = " 0123456789-MCHGISOENDRAV$\n\rcnos[]()=#:/\\@+.%*";
  long              total = 0, ingested = 0;
  for (int round = 0; round < std::atoi(argv[3]); ++round) {
    std::string text = base;
    const int   edits = round == 0 ? 0 : 1 + static_cast<int>(rng() % (sdf ? 400 : 4000));
    for (int e = 0; e < edits && !text.empty(); ++e) {
      const size_t pos = rng() % text.size();
      switch (rng() % 4) {
        case 0: text[pos] = alphabet[rng() % (sizeof(alphabet) - 1)]; break;
        case 1: text.insert(pos, 1, alphabet[rng() % (sizeof(alphabet) - 1)]); break;
        case 2: text.erase(pos, 1 + rng() % 3); break;
        default: text.erase(pos, rng() % (sdf ? 200 : 20)); break;
      }
    }
    if (round % 7 == 3 && !text.empty()) text.resize(rng() % text.size());  // a truncated file: no terminator after the last line
    char* buf = static_cast<char*>(std::malloc(text.empty() ? 1 : text.size()));
    std::memcpy(buf, text.data(), text.size());
    void*     h  = nullptr;
    const int rc = sdf ? nvmk_sdf_parse_text(buf, static_cast<int64_t>(text.size()), 4, round & 1, &h)
                       : nvmk_smiles_parse_text(buf, static_cast<int64_t>(text.size()), 4, round & 1, &h);
    if (rc != 0) return 1;
    int64_t n = 0;
    nvmk_smiles_size(h, &n);
    std::vector<int32_t> na(static_cast<size_t>(n) + 1), nb(static_cast<size_t>(n) + 1);
    std::vector<int8_t>  st(static_cast<size_t>(n) + 1);
    nvmk_smiles_counts(h, na.data(), nb.data(), st.data());
    for (int64_t i = 0; i < n; ++i) {
      std::vector<int32_t> a(6 * static_cast<size_t>(na[i]) + 1), b(4 * static_cast<size_t>(nb[i]) + 1);
      nvmk_smiles_graph(h, i, a.data(), b.data());
      ++total;
      if (st[i] == 0 && na[i] > 0 && na[i] < 1024 && nb[i] < 1024) {
        const int             slots = 1024;
        std::vector<uint32_t> ai(slots), bi(slots);
        std::vector<int16_t>  bx(slots * 8), bo(slots * 8);
        int16_t               count = 0;
        if (nvmk_smiles_morgan_inputs(h, &i, 1, slots, ai.data(), bi.data(), bx.data(), bo.data(), &count, 1) != 0) return 3;
        ++ingested;
      }
    }
    nvmk_smiles_free(h);
    std::free(buf);
  }
  std::printf("%s: %ld molecules, %ld ingested\n", argv[2], total, ingested);
  return 0;
}
