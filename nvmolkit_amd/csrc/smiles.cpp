// RDKit-free ingestion for the fingerprint path: SMILES -> molecular graph -> the input arrays of
// the Morgan kernel (MorganInvariantsGenerator::ComputeInvariantsInto, reference src/morgan_fingerprint_common.cpp:43-124).
//
// What the reference takes from RDKit for this path and how it is restated here (host code, no GPU involved):
//   * SMILES grammar (OpenSMILES): organic-subset and bracket atoms, isotopes, charges, explicit hydrogen counts, atom
//     classes (ignored), chirality marks (ignored: the reference fingerprints with includeChirality = false), bond
//     symbols - = # $ : / \, branches, ring closures (digits, %nn), dot-separated fragments.
//   * hydrogens written as atoms ([H]) are folded into their neighbour like RDKit's default removeHs (kept when they
//     carry an isotope or charge, bond to another hydrogen or have a degree other than one);
//   * implicit hydrogens of organic-subset atoms from RDKit's valence model (default valence lists below; aromatic atoms
//     count their aromatic bonds as 1.5 and take no hydrogen beyond the default valence);
//   * ring membership = the atom lies on a cycle (it has a bond that is not a bridge of the graph), which is what
//     RingInfo::numAtomRings(i) > 0 says for a cycle basis;
//   * AROMATICITY IS TAKEN FROM THE INPUT: lower-case atoms are aromatic and an unmarked ring bond between two of them is
//     an aromatic bond, as RDKit's canonical SMILES (ChEMBL, the reference's benchmarks/data/chembl_10k.smi) are written.
//     RDKit would additionally perceive aromaticity in Kekule-form input; a molecule with a Kekule-form ring that
//     satisfies Hueckel's rule is therefore REFUSED (status NVMK_SMILES_NEEDS_AROMATICITY) instead of being
//     fingerprinted with bond types RDKit would not use.  Valences RDKit's sanitisation rejects (or rewrites, like
//     five-valent nitro groups) are refused as well.
// Parity against RDKit cannot be pinned in this image; the independent Python restatement oracle/smiles.py, the
// element-count known answers of the reference's tests/test_morgan_fingerprint_ref.cpp:44-69 and hand-computed invariants
// are the checks (tests/test_smiles_ingestion.py).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace nvmk::smiles {
namespace {

constexpr int kMaxBondsPerAtom = 8;  // kMaxBondsPerAtom, src/morgan_fingerprint_common.h

enum Status : int8_t {
  kOk               = NVMK_SMILES_OK,
  kSyntax           = NVMK_SMILES_SYNTAX_ERROR,
  kValence          = NVMK_SMILES_VALENCE_ERROR,
  kNeedsAromaticity = NVMK_SMILES_NEEDS_AROMATICITY,
  kTooManyBonds     = NVMK_SMILES_TOO_MANY_BONDS,
};

// RDKit bond type values (Bond::BondType): what the Morgan bond invariant is (morgan_fingerprint_common.cpp:100)
constexpr uint8_t kSingle = 1, kDouble = 2, kTriple = 3, kQuadruple = 4, kAromatic = 12, kUnspecified = 0;

struct Atom {
  uint8_t  z        = 0;
  int8_t   charge   = 0;
  uint16_t isotope  = 0;
  int8_t   hExplicit = 0;  // hydrogen count written in the bracket (+ folded [H] atoms)
  int8_t   hImplicit = 0;
  bool     bracket  = false;
  bool     aromatic = false;
  bool     inRing   = false;
};
struct Bond {
  int     a = 0, b = 0;
  uint8_t order = kUnspecified;
  bool    ring  = false;
};
struct Graph {
  std::vector<Atom> atoms;
  std::vector<Bond> bonds;
  int8_t            status = kOk;
};

const char* const kSymbols[] = {
    "*",  "H",  "He", "Li", "Be", "B",  "C",  "N",  "O",  "F",  "Ne", "Na", "Mg", "Al", "Si", "P",  "S",  "Cl", "Ar", "K",
    "Ca", "Sc", "Ti", "V",  "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr", "Rb", "Sr", "Y",
    "Zr", "Nb", "Mo", "Tc", "Ru", "Rh", "Pd", "Ag", "Cd", "In", "Sn", "Sb", "Te", "I",  "Xe", "Cs", "Ba", "La", "Ce", "Pr",
    "Nd", "Pm", "Sm", "Eu", "Gd", "Tb", "Dy", "Ho", "Er", "Tm", "Yb", "Lu", "Hf", "Ta", "W",  "Re", "Os", "Ir", "Pt", "Au",
    "Hg", "Tl", "Pb", "Bi", "Po", "At", "Rn", "Fr", "Ra", "Ac", "Th", "Pa", "U",  "Np", "Pu", "Am", "Cm", "Bk", "Cf", "Es",
    "Fm", "Md", "No", "Lr", "Rf", "Db", "Sg", "Bh", "Hs", "Mt", "Ds", "Rg", "Cn", "Nh", "Fl", "Mc", "Lv", "Ts", "Og"};
constexpr int kNumElements = sizeof(kSymbols) / sizeof(kSymbols[0]);

// standard atomic weights (IUPAC abridged; mass number of the longest-lived isotope where there is none)
const double kWeights[kNumElements] = {
    0.0,     1.008,   4.0026,  6.94,    9.0122,  10.81,   12.011,  14.007,  15.999,  18.998,  20.180,  22.990,  24.305,  26.982,
    28.085,  30.974,  32.06,   35.45,   39.948,  39.098,  40.078,  44.956,  47.867,  50.942,  51.996,  54.938,  55.845,  58.933,
    58.693,  63.546,  65.38,   69.723,  72.630,  74.922,  78.971,  79.904,  83.798,  85.468,  87.62,   88.906,  91.224,  92.906,
    95.95,   98.0,    101.07,  102.91,  106.42,  107.87,  112.41,  114.82,  118.71,  121.76,  127.60,  126.90,  131.29,  132.91,
    137.33,  138.91,  140.12,  140.91,  144.24,  145.0,   150.36,  151.96,  157.25,  158.93,  162.50,  164.93,  167.26,  168.93,
    173.05,  174.97,  178.49,  180.95,  183.84,  186.21,  190.23,  192.22,  195.08,  196.97,  200.59,  204.38,  207.2,   208.98,
    209.0,   210.0,   222.0,   223.0,   226.0,   227.0,   232.04,  231.04,  238.03,  237.0,   244.0,   243.0,   247.0,   247.0,
    251.0,   252.0,   257.0,   258.0,   259.0,   262.0,   267.0,   268.0,   269.0,   270.0,   269.0,   278.0,   281.0,   282.0,
    285.0,   286.0,   289.0,   290.0,   293.0,   294.0,   294.0};

// Mass of a given isotope: exact values for the labels that occur in medicinal chemistry, the mass number otherwise (the
// mass defect is below 0.1 u, and the invariant only keeps int(mass - average weight)).
double isotope_mass(const int z, const int a) {
  struct Iso {
    int    z, a;
    double m;
  };
  static const Iso kIso[] = {{1, 1, 1.00783},    {1, 2, 2.01410},    {1, 3, 3.01605},    {6, 11, 11.01143},  {6, 12, 12.0},
                             {6, 13, 13.00335},  {6, 14, 14.00324},  {7, 13, 13.00574},  {7, 15, 15.00011},  {8, 15, 15.00307},
                             {8, 17, 16.99913},  {8, 18, 17.99916},  {9, 18, 18.00094},  {15, 32, 31.97391}, {15, 33, 32.97173},
                             {16, 34, 33.96787}, {16, 35, 34.96903}, {17, 36, 35.96831}, {17, 37, 36.96590}, {35, 76, 75.92454},
                             {35, 77, 76.92138}, {35, 82, 81.91680}, {53, 123, 122.90559}, {53, 124, 123.90621}, {53, 125, 124.90463},
                             {53, 131, 130.90612}};
  for (const Iso& i : kIso)
    if (i.z == z && i.a == a) return i.m;
  return static_cast<double>(a);
}

// default valence lists of the organic subset (RDKit's periodic table: the first entry is the default valence)
const int* valences_of(const int z, int& n) {
  static const int vB[] = {3}, vC[] = {4}, vN[] = {3}, vO[] = {2}, vF[] = {1}, vP[] = {3, 5, 7}, vS[] = {2, 4, 6}, vCl[] = {1}, vBr[] = {1},
                   vI[] = {1, 3, 5};
  switch (z) {
    case 5: n = 1; return vB;
    case 6: n = 1; return vC;
    case 7: n = 1; return vN;
    case 8: n = 1; return vO;
    case 9: n = 1; return vF;
    case 15: n = 3; return vP;
    case 16: n = 3; return vS;
    case 17: n = 1; return vCl;
    case 35: n = 1; return vBr;
    case 53: n = 3; return vI;
    default: n = 0; return nullptr;
  }
}

int element_of(const char* s, const int len) {
  for (int z = 0; z < kNumElements; ++z)
    if (static_cast<int>(std::strlen(kSymbols[z])) == len && std::strncmp(kSymbols[z], s, static_cast<size_t>(len)) == 0) return z;
  return -1;
}

struct Parser {
  const char* s;
  int         pos = 0;
  Graph&      g;
  explicit Parser(const char* str, Graph& graph) : s(str), g(graph) {}

  bool fail() {
    g.status = kSyntax;
    return false;
  }

  // "[" isotope? symbol chiral? hcount? charge? class? "]"
  bool bracket_atom(Atom& a) {
    ++pos;  // '['
    a.bracket = true;
    int iso = 0;
    bool hasIso = false;
    while (s[pos] >= '0' && s[pos] <= '9') {
      iso = iso * 10 + (s[pos++] - '0');
      hasIso = true;
      if (iso > 999) return fail();
    }
    a.isotope = static_cast<uint16_t>(hasIso ? iso : 0);
    if (s[pos] == '*') {
      a.z = 0;
      ++pos;
    } else if (s[pos] >= 'a' && s[pos] <= 'z') {  // aromatic symbols: b c n o p s se as te si
      int len = (s[pos + 1] >= 'a' && s[pos + 1] <= 'z' && ((s[pos] == 's' && (s[pos + 1] == 'e' || s[pos + 1] == 'i')) ||
                                                             (s[pos] == 'a' && s[pos + 1] == 's') || (s[pos] == 't' && s[pos + 1] == 'e')))
                    ? 2
                    : 1;
      char up[3] = {static_cast<char>(s[pos] - 32), len == 2 ? s[pos + 1] : '\0', '\0'};
      const int z = element_of(up, len);
      if (z != 5 && z != 6 && z != 7 && z != 8 && z != 15 && z != 16 && z != 34 && z != 33 && z != 52 && z != 14) return fail();
      a.z        = static_cast<uint8_t>(z);
      a.aromatic = true;
      pos += len;
    } else if (s[pos] >= 'A' && s[pos] <= 'Z') {
      int z = -1, len = 0;
      if (s[pos + 1] >= 'a' && s[pos + 1] <= 'z') {
        z   = element_of(s + pos, 2);
        len = 2;
      }
      if (z < 0) {
        z   = element_of(s + pos, 1);
        len = 1;
      }
      if (z < 0) return fail();
      a.z = static_cast<uint8_t>(z);
      pos += len;
    } else {
      return fail();
    }
    if (s[pos] == '@') {  // chirality: @, @@, or a class @TH1 @AL2 @SP3 @TB20 @OH30 — read and dropped
      ++pos;
      if (s[pos] == '@') {
        ++pos;
      } else if ((s[pos] == 'T' && (s[pos + 1] == 'H' || s[pos + 1] == 'B')) || (s[pos] == 'A' && s[pos + 1] == 'L') ||
                 (s[pos] == 'S' && s[pos + 1] == 'P') || (s[pos] == 'O' && s[pos + 1] == 'H')) {
        if (!(s[pos + 2] >= '0' && s[pos + 2] <= '9')) return fail();
        pos += 2;
        while (s[pos] >= '0' && s[pos] <= '9') ++pos;
      }
    }
    if (s[pos] == 'H') {
      ++pos;
      int h = 1;
      if (s[pos] >= '0' && s[pos] <= '9') h = s[pos++] - '0';
      a.hExplicit = static_cast<int8_t>(h);
    }
    if (s[pos] == '+' || s[pos] == '-') {
      const char sign = s[pos];
      int        q    = 0;
      while (s[pos] == sign) {
        ++q;
        ++pos;
      }
      if (q == 1 && s[pos] >= '0' && s[pos] <= '9') {
        q = 0;
        while (s[pos] >= '0' && s[pos] <= '9') q = q * 10 + (s[pos++] - '0');
      }
      if (q > 15) return fail();
      a.charge = static_cast<int8_t>(sign == '+' ? q : -q);
    }
    if (s[pos] == ':') {  // atom class
      ++pos;
      if (!(s[pos] >= '0' && s[pos] <= '9')) return fail();
      while (s[pos] >= '0' && s[pos] <= '9') ++pos;
    }
    if (s[pos] != ']') return fail();
    ++pos;
    return true;
  }

  bool organic_atom(Atom& a) {
    const char c = s[pos];
    if (c == '*') {
      a.z = 0;
      ++pos;
      return true;
    }
    if (c == 'C' && s[pos + 1] == 'l') {
      a.z = 17;
      pos += 2;
      return true;
    }
    if (c == 'B' && s[pos + 1] == 'r') {
      a.z = 35;
      pos += 2;
      return true;
    }
    switch (c) {
      case 'B': a.z = 5; break;
      case 'C': a.z = 6; break;
      case 'N': a.z = 7; break;
      case 'O': a.z = 8; break;
      case 'F': a.z = 9; break;
      case 'P': a.z = 15; break;
      case 'S': a.z = 16; break;
      case 'I': a.z = 53; break;
      case 'b': a.z = 5, a.aromatic = true; break;
      case 'c': a.z = 6, a.aromatic = true; break;
      case 'n': a.z = 7, a.aromatic = true; break;
      case 'o': a.z = 8, a.aromatic = true; break;
      case 'p': a.z = 15, a.aromatic = true; break;
      case 's': a.z = 16, a.aromatic = true; break;
      default: return fail();
    }
    ++pos;
    return true;
  }

  static uint8_t bond_of(const char c) {
    switch (c) {
      case '-': case '/': case '\\': return kSingle;
      case '=': return kDouble;
      case '#': return kTriple;
      case '$': return kQuadruple;
      case ':': return kAromatic;
      default: return kUnspecified;
    }
  }

  bool run() {
    struct Open {
      int     atom  = -1;
      uint8_t order = kUnspecified;
    };
    Open             ring[100];
    std::vector<int> stack;
    int              prev    = -1;
    uint8_t          pending = kUnspecified;
    bool             havePending = false;
    auto             add_bond = [&](const int a, const int b, const uint8_t order) {
      if (a == b) return false;
      for (const Bond& bd : g.bonds)
        if ((bd.a == a && bd.b == b) || (bd.a == b && bd.b == a)) return false;  // a second bond between the same atoms
      Bond bd;
      bd.a     = a;
      bd.b     = b;
      bd.order = order;
      g.bonds.push_back(bd);
      return true;
    };
    // the duplicate-bond scan above is O(bonds) per bond; bonds per molecule are few hundred at most, and only ring
    // closures can duplicate, so scan just for those
    auto add_chain_bond = [&](const int a, const int b, const uint8_t order) {
      Bond bd;
      bd.a     = a;
      bd.b     = b;
      bd.order = order;
      g.bonds.push_back(bd);
    };
    while (s[pos] == ' ' || s[pos] == '\t') ++pos;  // leading blanks; the SMILES ends at the next blank (name columns follow)
    while (s[pos] != '\0' && s[pos] != ' ' && s[pos] != '\t' && s[pos] != '\n' && s[pos] != '\r') {
      const char c = s[pos];
      if (c == '(') {
        if (prev < 0 || havePending) return fail();
        stack.push_back(prev);
        ++pos;
      } else if (c == ')') {
        if (stack.empty() || havePending) return fail();
        prev = stack.back();
        stack.pop_back();
        ++pos;
      } else if (c == '.') {
        if (havePending) return fail();
        prev = -1;
        ++pos;
      } else if (bond_of(c) != kUnspecified) {
        if (prev < 0 || havePending) return fail();
        pending     = bond_of(c);
        havePending = true;
        ++pos;
      } else if ((c >= '0' && c <= '9') || c == '%') {
        int label;
        if (c == '%') {
          if (!(s[pos + 1] >= '0' && s[pos + 1] <= '9' && s[pos + 2] >= '0' && s[pos + 2] <= '9')) return fail();
          label = (s[pos + 1] - '0') * 10 + (s[pos + 2] - '0');
          pos += 3;
        } else {
          label = c - '0';
          ++pos;
        }
        if (prev < 0) return fail();
        if (ring[label].atom < 0) {
          ring[label].atom  = prev;
          ring[label].order = havePending ? pending : kUnspecified;
        } else {
          uint8_t order = havePending ? pending : ring[label].order;
          if (havePending && ring[label].order != kUnspecified && ring[label].order != pending) return fail();
          if (!add_bond(ring[label].atom, prev, order)) return fail();
          ring[label].atom = -1;
        }
        havePending = false;
      } else {
        Atom a;
        if (c == '[') {
          if (!bracket_atom(a)) return false;
        } else {
          if (!organic_atom(a)) return false;
        }
        g.atoms.push_back(a);
        const int idx = static_cast<int>(g.atoms.size()) - 1;
        if (prev >= 0) add_chain_bond(prev, idx, havePending ? pending : kUnspecified);
        havePending = false;
        prev        = idx;
      }
    }
    if (havePending || !stack.empty()) return fail();
    for (const Open& o : ring)
      if (o.atom >= 0) return fail();
    return true;
  }
};

// bonds that lie on a cycle (= are not bridges): iterative depth-first search with low-links
void mark_ring_bonds(Graph& g) {
  const int n = static_cast<int>(g.atoms.size()), m = static_cast<int>(g.bonds.size());
  std::vector<int> head(static_cast<size_t>(n) + 1, 0), adjBond(static_cast<size_t>(2 * m)), adjAtom(static_cast<size_t>(2 * m));
  for (const Bond& b : g.bonds) {
    ++head[static_cast<size_t>(b.a) + 1];
    ++head[static_cast<size_t>(b.b) + 1];
  }
  for (int i = 0; i < n; ++i) head[static_cast<size_t>(i) + 1] += head[static_cast<size_t>(i)];
  std::vector<int> fill(head.begin(), head.end() - 1);
  for (int k = 0; k < m; ++k) {
    const Bond& b = g.bonds[static_cast<size_t>(k)];
    adjBond[static_cast<size_t>(fill[static_cast<size_t>(b.a)])]   = k;
    adjAtom[static_cast<size_t>(fill[static_cast<size_t>(b.a)]++)] = b.b;
    adjBond[static_cast<size_t>(fill[static_cast<size_t>(b.b)])]   = k;
    adjAtom[static_cast<size_t>(fill[static_cast<size_t>(b.b)]++)] = b.a;
  }
  std::vector<int> disc(static_cast<size_t>(n), -1), low(static_cast<size_t>(n), 0), parentBond(static_cast<size_t>(n), -1),
      it(static_cast<size_t>(n), 0), stack;
  int timer = 0;
  for (int root = 0; root < n; ++root) {
    if (disc[static_cast<size_t>(root)] >= 0) continue;
    disc[static_cast<size_t>(root)] = low[static_cast<size_t>(root)] = timer++;
    it[static_cast<size_t>(root)]   = head[static_cast<size_t>(root)];
    stack.push_back(root);
    while (!stack.empty()) {
      const int u = stack.back();
      if (it[static_cast<size_t>(u)] < head[static_cast<size_t>(u) + 1]) {
        const int e = it[static_cast<size_t>(u)]++;
        const int k = adjBond[static_cast<size_t>(e)], v = adjAtom[static_cast<size_t>(e)];
        if (k == parentBond[static_cast<size_t>(u)]) continue;
        if (disc[static_cast<size_t>(v)] >= 0) {
          low[static_cast<size_t>(u)] = std::min(low[static_cast<size_t>(u)], disc[static_cast<size_t>(v)]);
        } else {
          disc[static_cast<size_t>(v)] = low[static_cast<size_t>(v)] = timer++;
          parentBond[static_cast<size_t>(v)]                          = k;
          it[static_cast<size_t>(v)]                                  = head[static_cast<size_t>(v)];
          stack.push_back(v);
        }
      } else {
        stack.pop_back();
        const int k = parentBond[static_cast<size_t>(u)];
        if (k >= 0) {
          const Bond& b = g.bonds[static_cast<size_t>(k)];
          const int   p = b.a == u ? b.b : b.a;
          low[static_cast<size_t>(p)] = std::min(low[static_cast<size_t>(p)], low[static_cast<size_t>(u)]);
          // the tree bond (p, u) is a bridge iff nothing below u reaches p or above
          g.bonds[static_cast<size_t>(k)].ring = low[static_cast<size_t>(u)] <= disc[static_cast<size_t>(p)];
        }
      }
    }
  }
  // back edges close cycles by definition
  for (int k = 0; k < m; ++k) {
    Bond& b = g.bonds[static_cast<size_t>(k)];
    if (parentBond[static_cast<size_t>(b.a)] != k && parentBond[static_cast<size_t>(b.b)] != k) b.ring = true;
  }
  for (const Bond& b : g.bonds)
    if (b.ring) g.atoms[static_cast<size_t>(b.a)].inRing = g.atoms[static_cast<size_t>(b.b)].inRing = true;
}

// RDKit's default removeHs on what a SMILES can express: a hydrogen atom is folded into its neighbour unless it is
// labelled (isotope), charged, not singly bonded to exactly one non-hydrogen atom.
void fold_hydrogens(Graph& g) {
  const int        n = static_cast<int>(g.atoms.size());
  std::vector<int> degree(static_cast<size_t>(n), 0), onlyBond(static_cast<size_t>(n), -1);
  for (size_t k = 0; k < g.bonds.size(); ++k) {
    ++degree[static_cast<size_t>(g.bonds[k].a)];
    ++degree[static_cast<size_t>(g.bonds[k].b)];
    onlyBond[static_cast<size_t>(g.bonds[k].a)] = onlyBond[static_cast<size_t>(g.bonds[k].b)] = static_cast<int>(k);
  }
  std::vector<char> drop(static_cast<size_t>(n), 0);
  bool              any = false;
  for (int i = 0; i < n; ++i) {
    const Atom& a = g.atoms[static_cast<size_t>(i)];
    if (a.z != 1 || a.isotope != 0 || a.charge != 0 || a.hExplicit != 0 || degree[static_cast<size_t>(i)] != 1) continue;
    const Bond& b = g.bonds[static_cast<size_t>(onlyBond[static_cast<size_t>(i)])];
    const int   o = b.a == i ? b.b : b.a;
    if (g.atoms[static_cast<size_t>(o)].z == 1 || (b.order != kSingle && b.order != kUnspecified)) continue;
    drop[static_cast<size_t>(i)] = 1;
    any                          = true;
    if (g.atoms[static_cast<size_t>(o)].bracket) ++g.atoms[static_cast<size_t>(o)].hExplicit;  // organic-subset atoms recount below
  }
  if (!any) return;
  std::vector<int> renum(static_cast<size_t>(n), -1);
  std::vector<Atom> atoms;
  for (int i = 0; i < n; ++i)
    if (!drop[static_cast<size_t>(i)]) {
      renum[static_cast<size_t>(i)] = static_cast<int>(atoms.size());
      atoms.push_back(g.atoms[static_cast<size_t>(i)]);
    }
  std::vector<Bond> bonds;
  for (const Bond& b : g.bonds)
    if (!drop[static_cast<size_t>(b.a)] && !drop[static_cast<size_t>(b.b)]) {
      Bond nb = b;
      nb.a    = renum[static_cast<size_t>(b.a)];
      nb.b    = renum[static_cast<size_t>(b.b)];
      bonds.push_back(nb);
    }
  g.atoms.swap(atoms);
  g.bonds.swap(bonds);
}

// twice the valence contribution of a bond (RDKit counts an aromatic bond as 1.5)
int half_orders(const uint8_t order) {
  switch (order) {
    case kDouble: return 4;
    case kTriple: return 6;
    case kQuadruple: return 8;
    case kAromatic: return 3;
    default: return 2;
  }
}

// Atom::calcExplicitValence / calcImplicitValence of RDKit for atoms written without brackets
bool assign_implicit_hydrogens(Graph& g) {
  std::vector<int> sum2(g.atoms.size(), 0);
  for (const Bond& b : g.bonds) {
    sum2[static_cast<size_t>(b.a)] += half_orders(b.order);
    sum2[static_cast<size_t>(b.b)] += half_orders(b.order);
  }
  for (size_t i = 0; i < g.atoms.size(); ++i) {
    Atom& a = g.atoms[i];
    if (a.bracket || a.z == 0) continue;
    int        nv = 0;
    const int* v  = valences_of(a.z, nv);
    if (v == nullptr) return false;
    double accum = 0.5 * sum2[i];
    if (a.aromatic) {
      const int dv = v[0];
      if (accum > dv) {  // no hydrogen can be added: the closest allowed valence not above the bond-order sum
        int pval = dv;
        for (int k = 0; k < nv; ++k) {
          if (v[k] > accum) break;
          pval = v[k];
        }
        accum = pval;
      }
      const int ev = static_cast<int>(std::lround(accum + 0.1));
      a.hImplicit  = static_cast<int8_t>(ev <= dv ? dv - ev : 0);
    } else {
      const int ev    = static_cast<int>(std::lround(accum + 0.1));
      int       found = -1;
      for (int k = 0; k < nv; ++k)
        if (v[k] >= ev) {
          found = v[k];
          break;
        }
      if (found < 0) return false;  // RDKit: "Explicit valence ... is greater than permitted"
      a.hImplicit = static_cast<int8_t>(found - ev);
    }
  }
  return true;
}

// Would RDKit perceive an aromatic ring in the Kekule-form part of this molecule?  Conservative single-ring test:
// the smallest ring through every non-aromatic ring double bond is examined; it counts as Hueckel-aromatic when every
// member is sp2-like (a ring double bond, or a heteroatom / anion that donates a lone pair, or a carbon whose
// exocyclic double bond goes to an electronegative atom and contributes no electron) and the electrons sum to 4k + 2.
bool kekule_ring_looks_aromatic(const Graph& g) {
  const int n = static_cast<int>(g.atoms.size());
  std::vector<std::vector<std::pair<int, int>>> adj(static_cast<size_t>(n));  // (neighbour, bond)
  for (size_t k = 0; k < g.bonds.size(); ++k) {
    adj[static_cast<size_t>(g.bonds[k].a)].push_back({g.bonds[k].b, static_cast<int>(k)});
    adj[static_cast<size_t>(g.bonds[k].b)].push_back({g.bonds[k].a, static_cast<int>(k)});
  }
  for (size_t k0 = 0; k0 < g.bonds.size(); ++k0) {
    const Bond& b0 = g.bonds[k0];
    if (!b0.ring || b0.order != kDouble) continue;
    // shortest path from b0.a to b0.b that avoids b0, over ring bonds: with b0 it is the smallest ring through b0
    std::vector<int> from(static_cast<size_t>(n), -2), queue;
    from[static_cast<size_t>(b0.a)] = -1;
    queue.push_back(b0.a);
    for (size_t q = 0; q < queue.size() && from[static_cast<size_t>(b0.b)] == -2; ++q) {
      const int u = queue[q];
      for (const auto& [v, k] : adj[static_cast<size_t>(u)]) {
        if (static_cast<size_t>(k) == k0 || !g.bonds[static_cast<size_t>(k)].ring || from[static_cast<size_t>(v)] != -2) continue;
        from[static_cast<size_t>(v)] = u;
        queue.push_back(v);
      }
    }
    if (from[static_cast<size_t>(b0.b)] == -2) continue;
    std::vector<int> ringAtoms;
    for (int v = b0.b; v != -1; v = from[static_cast<size_t>(v)]) ringAtoms.push_back(v);
    if (ringAtoms.size() > 8) continue;
    std::vector<char> inThis(static_cast<size_t>(n), 0);
    for (const int v : ringAtoms) inThis[static_cast<size_t>(v)] = 1;
    int  electrons = 0;
    bool conjugated = true;
    for (const int v : ringAtoms) {
      const Atom& a            = g.atoms[static_cast<size_t>(v)];
      bool        ringDouble   = false, exoDouble = false, exoToHetero = false;
      int         nBonds       = 0;
      for (const auto& [w, k] : adj[static_cast<size_t>(v)]) {
        ++nBonds;
        const uint8_t o = g.bonds[static_cast<size_t>(k)].order;
        if (o == kDouble) {
          if (inThis[static_cast<size_t>(w)]) {
            ringDouble = true;
          } else {
            exoDouble   = true;
            const int zw = g.atoms[static_cast<size_t>(w)].z;
            exoToHetero  = zw == 7 || zw == 8 || zw == 16;
          }
        }
        if (o == kTriple || o == kAromatic) conjugated = false;  // aromatic-form rings are the input's business
      }
      // RDKit's isAtomCandForArom: main-group ring atoms with at most three neighbours (hydrogens included)
      const bool element = a.z == 5 || a.z == 6 || a.z == 7 || a.z == 8 || a.z == 15 || a.z == 16 || a.z == 33 || a.z == 34 || a.z == 52;
      if (!element || nBonds + a.hExplicit + a.hImplicit > 3) conjugated = false;
      if (!conjugated) break;
      if (ringDouble) {
        electrons += 1;
      } else if (exoDouble) {
        if (!exoToHetero) conjugated = false;  // exocyclic C=C: RDKit does not count the ring as aromatic
      } else if ((a.z == 7 || a.z == 15) && a.charge == 0 && nBonds + a.hExplicit + a.hImplicit == 3) {
        electrons += 2;
      } else if ((a.z == 8 || a.z == 16 || a.z == 34) && a.charge == 0 && nBonds == 2) {
        electrons += 2;
      } else if (a.z == 6 && a.charge == -1) {
        electrons += 2;
      } else if ((a.z == 6 && a.charge == 1) || (a.z == 5 && a.charge == 0 && nBonds + a.hExplicit + a.hImplicit == 3)) {
        electrons += 0;
      } else {
        conjugated = false;
      }
      if (!conjugated) break;
    }
    if (conjugated && electrons >= 2 && (electrons - 2) % 4 == 0) return true;
  }
  return false;
}

void build(const char* s, Graph& g) {
  g = Graph();
  if (s == nullptr) {
    g.status = kSyntax;
    return;
  }
  Parser p(s, g);
  if (!p.run()) {
    g.atoms.clear();
    g.bonds.clear();
    if (g.status == kOk) g.status = kSyntax;
    return;
  }
  fold_hydrogens(g);
  mark_ring_bonds(g);
  for (Bond& b : g.bonds) {
    if (b.order != kUnspecified) continue;
    const bool arom = g.atoms[static_cast<size_t>(b.a)].aromatic && g.atoms[static_cast<size_t>(b.b)].aromatic && b.ring;
    b.order         = arom ? kAromatic : kSingle;
  }
  if (!assign_implicit_hydrogens(g)) {
    g.status = kValence;
    return;
  }
  std::vector<int> degree(g.atoms.size(), 0);
  for (const Bond& b : g.bonds) {
    if (++degree[static_cast<size_t>(b.a)] > kMaxBondsPerAtom || ++degree[static_cast<size_t>(b.b)] > kMaxBondsPerAtom) {
      g.status = kTooManyBonds;
      return;
    }
  }
  if (kekule_ring_looks_aromatic(g)) g.status = kNeedsAromaticity;
}

uint32_t hash_combine(const uint32_t seed, const uint32_t v) { return seed ^ (v + 0x9e3779b9u + (seed << 6) + (seed >> 2)); }

}  // namespace

struct Set {
  std::vector<Graph> graphs;
};

template <typename F> void parallel_for(const int64_t n, int threads, F&& body) {
  if (threads <= 0) threads = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
  threads = static_cast<int>(std::min<int64_t>(threads, std::max<int64_t>(1, n / 256)));  // a thread is worth starting for >= 256 molecules
  if (threads <= 1) {
    for (int64_t i = 0; i < n; ++i) body(i);
    return;
  }
  std::atomic<int64_t>     next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&] {
      for (;;) {
        const int64_t lo = next.fetch_add(64);
        if (lo >= n) return;
        for (int64_t i = lo; i < std::min(n, lo + 64); ++i) body(i);
      }
    });
  for (std::thread& t : pool) t.join();
}

}  // namespace nvmk::smiles

extern "C" {

int nvmk_smiles_parse(const char* const* smiles, const int64_t n_mols, const int n_threads, void** handle) {
  NVMK_REQUIRE(handle != nullptr && (smiles != nullptr || n_mols == 0) && n_mols >= 0, "nvmk_smiles_parse: NULL argument or negative count");
  auto set = std::make_unique<nvmk::smiles::Set>();
  set->graphs.resize(static_cast<size_t>(n_mols));
  nvmk::smiles::parallel_for(n_mols, n_threads, [&](const int64_t i) { nvmk::smiles::build(smiles[i], set->graphs[static_cast<size_t>(i)]); });
  *handle = set.release();
  return NVMK_OK;
}

int nvmk_smiles_free(void* handle) {
  delete static_cast<nvmk::smiles::Set*>(handle);
  return NVMK_OK;
}

int nvmk_smiles_counts(const void* handle, int32_t* n_atoms, int32_t* n_bonds, int8_t* status) {
  NVMK_REQUIRE(handle != nullptr, "nvmk_smiles_counts: NULL handle");
  const auto& graphs = static_cast<const nvmk::smiles::Set*>(handle)->graphs;
  for (size_t i = 0; i < graphs.size(); ++i) {
    if (n_atoms != nullptr) n_atoms[i] = static_cast<int32_t>(graphs[i].atoms.size());
    if (n_bonds != nullptr) n_bonds[i] = static_cast<int32_t>(graphs[i].bonds.size());
    if (status != nullptr) status[i] = graphs[i].status;
  }
  return NVMK_OK;
}

int nvmk_smiles_graph(const void* handle, const int64_t mol, int32_t* atom_fields, int32_t* bond_fields) {
  NVMK_REQUIRE(handle != nullptr, "nvmk_smiles_graph: NULL handle");
  const auto& graphs = static_cast<const nvmk::smiles::Set*>(handle)->graphs;
  NVMK_REQUIRE(mol >= 0 && static_cast<size_t>(mol) < graphs.size(), "nvmk_smiles_graph: molecule index out of range");
  const auto& g = graphs[static_cast<size_t>(mol)];
  for (size_t i = 0; atom_fields != nullptr && i < g.atoms.size(); ++i) {
    const auto& a          = g.atoms[i];
    atom_fields[6 * i + 0] = a.z;
    atom_fields[6 * i + 1] = a.charge;
    atom_fields[6 * i + 2] = a.isotope;
    atom_fields[6 * i + 3] = a.hExplicit + a.hImplicit;
    atom_fields[6 * i + 4] = a.aromatic ? 1 : 0;
    atom_fields[6 * i + 5] = a.inRing ? 1 : 0;
  }
  for (size_t k = 0; bond_fields != nullptr && k < g.bonds.size(); ++k) {
    bond_fields[4 * k + 0] = g.bonds[k].a;
    bond_fields[4 * k + 1] = g.bonds[k].b;
    bond_fields[4 * k + 2] = g.bonds[k].order;
    bond_fields[4 * k + 3] = g.bonds[k].ring ? 1 : 0;
  }
  return NVMK_OK;
}

int nvmk_smiles_morgan_inputs(const void* handle, const int64_t* mol_ids, const int64_t n_sel, const int max_atoms, uint32_t* atom_inv,
                              uint32_t* bond_inv, int16_t* bond_idx, int16_t* bond_other, int16_t* n_atoms, const int n_threads) {
  using namespace nvmk::smiles;
  NVMK_REQUIRE(handle != nullptr && atom_inv != nullptr && bond_inv != nullptr && bond_idx != nullptr && bond_other != nullptr &&
                   n_atoms != nullptr && n_sel >= 0,
               "nvmk_smiles_morgan_inputs: NULL argument");
  NVMK_REQUIRE(max_atoms > 0 && max_atoms <= 32767, "nvmk_smiles_morgan_inputs: max_atoms out of range");
  const auto& graphs = static_cast<const Set*>(handle)->graphs;
  for (int64_t s = 0; s < n_sel; ++s) {
    const int64_t m = mol_ids != nullptr ? mol_ids[s] : s;
    NVMK_REQUIRE(m >= 0 && static_cast<size_t>(m) < graphs.size(), "nvmk_smiles_morgan_inputs: molecule index %lld out of range",
                 static_cast<long long>(m));
    const Graph& g = graphs[static_cast<size_t>(m)];
    NVMK_REQUIRE(g.status == kOk, "nvmk_smiles_morgan_inputs: molecule %lld was not ingested (status %d)", static_cast<long long>(m), g.status);
    NVMK_REQUIRE(static_cast<int>(g.atoms.size()) < max_atoms && static_cast<int>(g.bonds.size()) < max_atoms,
                 "nvmk_smiles_morgan_inputs: molecule %lld does not fit a %d-atom bucket", static_cast<long long>(m), max_atoms);
  }
  const size_t stride = static_cast<size_t>(max_atoms);
  parallel_for(n_sel, n_threads, [&](const int64_t s) {
    const Graph& g   = graphs[static_cast<size_t>(mol_ids != nullptr ? mol_ids[s] : s)];
    uint32_t*    ai  = atom_inv + static_cast<size_t>(s) * stride;
    uint32_t*    bi  = bond_inv + static_cast<size_t>(s) * stride;
    int16_t*     bix = bond_idx + static_cast<size_t>(s) * stride * kMaxBondsPerAtom;
    int16_t*     bo  = bond_other + static_cast<size_t>(s) * stride * kMaxBondsPerAtom;
    std::fill(ai, ai + stride, 0u);
    std::fill(bi, bi + stride, 0u);
    std::fill(bix, bix + stride * kMaxBondsPerAtom, static_cast<int16_t>(-1));
    std::fill(bo, bo + stride * kMaxBondsPerAtom, static_cast<int16_t>(-1));
    const int na = static_cast<int>(g.atoms.size());
    n_atoms[s]   = static_cast<int16_t>(na);
    std::vector<int> deg(static_cast<size_t>(na), 0), nbrH(static_cast<size_t>(na), 0);
    for (size_t k = 0; k < g.bonds.size(); ++k) {
      const Bond& b = g.bonds[k];
      bi[k]         = b.order;
      const int ends[2][2] = {{b.a, b.b}, {b.b, b.a}};
      for (const auto& e : ends) {
        const int slot = deg[static_cast<size_t>(e[0])]++;
        bix[static_cast<size_t>(e[0]) * kMaxBondsPerAtom + static_cast<size_t>(slot)] = static_cast<int16_t>(k);
        bo[static_cast<size_t>(e[0]) * kMaxBondsPerAtom + static_cast<size_t>(slot)]  = static_cast<int16_t>(e[1]);
        if (g.atoms[static_cast<size_t>(e[1])].z == 1) ++nbrH[static_cast<size_t>(e[0])];
      }
    }
    for (int i = 0; i < na; ++i) {
      const Atom&    a  = g.atoms[static_cast<size_t>(i)];
      const int      hs = a.hExplicit + a.hImplicit;
      const double   mass = a.isotope != 0 ? isotope_mass(a.z, a.isotope) : kWeights[a.z];
      const uint32_t comps[5] = {a.z, static_cast<uint32_t>(hs + deg[static_cast<size_t>(i)]), static_cast<uint32_t>(hs + nbrH[static_cast<size_t>(i)]),
                                 static_cast<uint32_t>(static_cast<int32_t>(a.charge)),
                                 static_cast<uint32_t>(static_cast<int32_t>(mass - kWeights[a.z]))};
      uint32_t seed = 0;
      for (const uint32_t c : comps) seed = hash_combine(seed, c);
      if (a.inRing) seed = hash_combine(seed, 1u);
      ai[i] = seed;
    }
  });
  return NVMK_OK;
}

}  // extern "C"
