// RDKit-free ingestion for the fingerprint path: SMILES or SD-file records (MolfileReader near the
// end of this file; everything from the hydrogen folding on is shared) -> molecular graph -> the input arrays of
// the Morgan kernel (MorganInvariantsGenerator::ComputeInvariantsInto, reference src/morgan_fingerprint_common.cpp:43-124).
//
// What the reference takes from RDKit for this path and how it is restated here (host code, no GPU involved):
//   * SMILES grammar (OpenSMILES): organic-subset and bracket atoms, isotopes, charges, explicit hydrogen counts, atom
//     classes (ignored), chirality marks (ignored: the reference fingerprints with includeChirality = false), bond
//     symbols - = # $ : / \, branches, ring closures (digits, %nn), dot-separated fragments; and the two spellings RDKit's
//     parser reads beyond it: ring-closure labels %(n) up to five digits and atoms by atomic number, [#6].
//   * hydrogens written as atoms ([H]) are folded into their neighbour like RDKit's default removeHs (kept when they
//     carry an isotope or charge, bond to another hydrogen or have a degree other than one);
//   * implicit hydrogens of organic-subset atoms from RDKit's valence model (default valence lists below; aromatic atoms
//     count their aromatic bonds as 1.5 and take no hydrogen beyond the default valence);
//   * ring membership = the atom lies on a cycle (it has a bond that is not a bridge of the graph), which is what
//     RingInfo::numAtomRings(i) > 0 says for a cycle basis;
//   * AROMATICITY IS RDKIT'S, NOT THE INPUT'S.  Like RDKit's sanitisation, every molecule is first Kekulised (what the
//     input wrote in lower case gets alternating single / double bonds; no Kekule structure = refused, RDKit's "Can't
//     kekulize mol") and its aromaticity is then perceived from the Kekule structure with RDKit's default model
//     (perceive_aromaticity below: candidate rings are the shortest rings through each ring bond whose atoms can all
//     donate, electrons per atom from its unsaturation / lone pair / exocyclic double bond, 4n+2 over single rings and
//     over combinations of up to six rings fused through single shared bonds).  By default the outcome must EQUAL what the
//     input said - which is the case for SMILES RDKit wrote (all 10 000 of the reference's benchmark file) - otherwise the
//     molecule is refused (NVMK_SMILES_NEEDS_AROMATICITY: Kekule-form input, or aromatic input from another toolkit's
//     model) rather than fingerprinted with bond types RDKit would not use.  With NVMK_SMILES_PERCEIVE_AROMATICITY
//     (nvmk_smiles_parse_flags; what the Python classes pass unless told otherwise) the outcome is applied, as RDKit does.  The perception is checked on the
//     aromaticity RDKit itself recorded: every aromatic ChEMBL SMILES of tests/golden is Kekulised by the oracle, read
//     back here and comes out with exactly the aromatic atoms and bonds RDKit wrote - all 8864 aromatic molecules of the
//     10 000, porphyrins and fullerene adducts included (tests/test_smiles_aromaticity.py).
//   * the hypervalent spellings RDKit's sanitisation rewrites first (MolOps::cleanUp: N(=O)=O, N=N#N, C=P(=O)X, O=Cl(=O)O)
//     are rewritten the same way (clean_up below); valences it rejects are refused (NVMK_SMILES_VALENCE_ERROR).
// Host throughput: one call parses a whole text buffer (nvmk_smiles_parse_text) or an array of strings on all host threads;
// each thread works in its own Scratch (no allocation per molecule) and fills chunks of 512 molecules stored back to back.
// Parity against RDKit's parser cannot be pinned in this image; the independent Python restatement oracle/smiles.py, the
// element-count known answers of the reference's tests/test_morgan_fingerprint_ref.cpp:44-69, hand-computed invariants
// and the ChEMBL round trip above are the checks (tests/test_smiles_ingestion.py, tests/test_smiles_aromaticity.py).
#include <algorithm>
#include <array>
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
  kNoKekuleForm     = NVMK_SMILES_NO_KEKULE_FORM,
  kIsotope          = NVMK_SMILES_UNSUPPORTED_ISOTOPE,
};

// RDKit bond type values (Bond::BondType): what the Morgan bond invariant is (morgan_fingerprint_common.cpp:100)
constexpr uint8_t kSingle = 1, kDouble = 2, kTriple = 3, kQuadruple = 4, kAromatic = 12, kUnspecified = 0;
constexpr uint8_t kDirectional = 200;  // parser only: '/' or '\\' was written; stored as kUnspecified

struct Atom {
  uint8_t  z        = 0;
  int8_t   charge   = 0;
  uint16_t isotope  = 0;
  int8_t   hExplicit = 0;  // hydrogen count written in the bracket (+ folded [H] atoms)
  int8_t   hImplicit = 0;
  bool     bracket  = false;
  bool     aromatic = false;
  bool     inRing   = false;
  uint8_t  radical  = 0;  // unpaired electrons a molfile declares (M  RAD): they take the place of implicit hydrogens
};
struct Bond {
  int     a = 0, b = 0;
  uint8_t order = kUnspecified;
  bool    ring  = false;
  bool    dir   = false;  // written '/' or '\\' (kept for RDKit's removeHs rule on stereo-defining hydrogens)
};
struct Graph {
  std::vector<Atom> atoms;
  std::vector<Bond> bonds;
  int8_t            status = kOk;
};

// Per-thread working storage of one molecule at a time: every array a pass needs lives here and keeps its capacity from
// molecule to molecule, so that parsing allocates nothing once a thread has seen its largest molecule (a million small
// vectors per call was what the parser spent most of its time on, and what kept it from scaling over threads).
struct Scratch {
  Graph            g;
  std::vector<int> branchStack;                     // parser
  std::vector<int> head, adjBond, adjAtom, fill;    // adjacency in CSR form: neighbours of i are [head[i], head[i + 1])
  std::vector<int> disc, low, parentBond, it, dfs;  // bridge search
  std::vector<int> degree, onlyBond, renum, sum2;   // hydrogen folding, valences
  std::vector<char> drop;
  std::vector<Atom> keptAtoms;
  std::vector<Bond> keptBonds;
  std::vector<int> donated, from, depth, queue, ring;  // aromaticity
  // candidate rings and fused systems of the perception, all flat: ring r = ringAtoms[ringStart[r] .. ringStart[r + 1]) in
  // cycle order, its bonds (sorted) at the same positions of ringBonds; shortest_rings_through appends to found*
  std::vector<int> ringStart, ringAtoms, ringBonds, foundStart, foundAtoms, fusedStart, fusedList, fusedFill;
  std::vector<int> ringCount, bondCount, systemOf, members, ringStack, pathBonds;
  std::vector<char> aromAtom, aromBond, bondSeen;
  std::vector<std::array<int, 6>> level, next;
  std::vector<int> sigma, partner, free_;           // Kekule structure of the aromatic part
  std::vector<char> needs, inputAromatic;
  std::vector<uint8_t> inputOrder;
  std::vector<unsigned> seen;                       // visit stamps of the ring search
  unsigned         stamp = 0;
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

// Mass excess (mass - mass number, in u) of the nuclides of mass number a that chemistry meets: +0.015 for the lightest,
// falling to -0.1 around a = 120 (tin), back through 0 near a = 214 and up to +0.08 at californium; the band around that
// curve covers every isotope within a few neutrons of stability (2H +0.014, 14C +0.003, 32P -0.026, 56Fe -0.065,
// 99Tc -0.094, 120Sn -0.098, 131I -0.094, 180Hf -0.053, 208Pb -0.023, 223Ra +0.019, 238U +0.051, 252Cf +0.082).
void mass_excess_window(const int a, double& lo, double& hi) {
  double c, w;
  if (a <= 12) {
    c = 0.02, w = 0.025;
  } else if (a <= 40) {
    c = 0.015 - (a - 12) * 0.00186, w = 0.02;
  } else if (a <= 200) {
    const double x = (a - 120) / 100.0;
    c = -0.1 * (1.0 - x * x), w = 0.04;
  } else {
    c = -0.03 + (a - 200) * 0.0021, w = a <= 260 ? 0.03 : 0.08;
  }
  lo = c - w;
  hi = c + w;
}

// Mass of a given isotope: exact values for the labels that occur in medicinal chemistry, otherwise the mass number plus
// the centre of the band of mass excesses at that mass number — used only where the whole band gives the same
// int(mass - average weight) (isotope_known below).
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
                             {53, 131, 130.90612}, {27, 57, 56.93629},   {71, 177, 176.94376}};
  if (z <= 0 || z >= kNumElements) return static_cast<double>(a);  // dummy atoms have no nuclide: the label IS the mass (RDKit: isotope as a double)
  for (const Iso& i : kIso)
    if (i.z == z && i.a == a) return i.m;
  double lo, hi;
  mass_excess_window(a, lo, hi);
  return static_cast<double>(a) + 0.5 * (lo + hi);
}
// The invariant keeps int(mass - average weight).  For an isotope outside the table the mass is only known to lie in the band
// of mass excesses of the nuclides near the valley of stability (mass_excess_window); the label is taken when every mass in
// that band truncates to the same integer, and refused otherwise ([177Lu], [211At], [223Ra] ...) rather than risk an
// invariant RDKit's exact isotope table would not give.
bool isotope_known(const int z, const int a) {
  if (z <= 0 || z >= kNumElements) return true;  // dummy atoms: weight 0, the label is the mass
  double lo, hi;
  mass_excess_window(a, lo, hi);
  const double mid = static_cast<double>(a) + 0.5 * (lo + hi);
  if (isotope_mass(z, a) != mid) return true;    // tabulated
  return static_cast<int32_t>(a + lo - kWeights[z]) == static_cast<int32_t>(a + hi - kWeights[z]);
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
    // not in the SMILES organic subset, but atoms of a molfile are all written without a hydrogen count
    case 1: n = 1; return vF;     // {1}
    case 14: n = 1; return vC;    // Si {4}
    case 33: n = 2; return vP;    // As {3, 5}
    case 34: n = 3; return vS;    // Se {2, 4, 6}
    case 52: n = 3; return vS;    // Te {2, 4, 6}
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
  std::vector<int>& stack;
  explicit Parser(const char* str, Scratch& sc) : s(str), g(sc.g), stack(sc.branchStack) { stack.clear(); }

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
    } else if (s[pos] == '#') {  // the element by atomic number, "[#6]" (RDKit's SMILES parser reads it; 0 = a dummy atom)
      ++pos;
      if (!(s[pos] >= '0' && s[pos] <= '9')) return fail();
      int z = 0;
      while (s[pos] >= '0' && s[pos] <= '9') {
        z = z * 10 + (s[pos++] - '0');
        if (z > 118) return fail();
      }
      a.z = static_cast<uint8_t>(z);
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
        while (s[pos] >= '0' && s[pos] <= '9') {
          q = q * 10 + (s[pos++] - '0');
          if (q > 15) return fail();  // (bounded inside the loop: no overflow on a long digit string)
        }
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
      case '-': return kSingle;
      case '/': case '\\': return kDirectional;  // a direction, not an order: aromatic between aromatic ring atoms, else single
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
    struct Far {  // ring-closure labels written %(n) with n >= 100: rare, kept in a short list
      int  label;
      Open open;
    };
    std::vector<Far> far;
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
      bd.order = order == kDirectional ? kUnspecified : order;
      bd.dir   = order == kDirectional;
      g.bonds.push_back(bd);
      return true;
    };
    // the duplicate-bond scan above is O(bonds) per bond; bonds per molecule are few hundred at most, and only ring
    // closures can duplicate, so scan just for those
    auto add_chain_bond = [&](const int a, const int b, const uint8_t order) {
      Bond bd;
      bd.a     = a;
      bd.b     = b;
      bd.order = order == kDirectional ? kUnspecified : order;
      bd.dir   = order == kDirectional;
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
        if (c == '%' && s[pos + 1] == '(') {  // %(n): RDKit's form for labels beyond 99, up to five digits
          pos += 2;
          if (!(s[pos] >= '0' && s[pos] <= '9')) return fail();
          label      = 0;
          int digits = 0;
          while (s[pos] >= '0' && s[pos] <= '9') {
            label = label * 10 + (s[pos++] - '0');
            if (++digits > 5) return fail();
          }
          if (s[pos] != ')') return fail();
          ++pos;
        } else if (c == '%') {
          if (!(s[pos + 1] >= '0' && s[pos + 1] <= '9' && s[pos + 2] >= '0' && s[pos + 2] <= '9')) return fail();
          label = (s[pos + 1] - '0') * 10 + (s[pos + 2] - '0');
          pos += 3;
        } else {
          label = c - '0';
          ++pos;
        }
        if (prev < 0) return fail();
        Open* slot = nullptr;
        if (label < 100) {
          slot = &ring[label];
        } else {
          for (Far& f : far)
            if (f.label == label) slot = &f.open;
          if (slot == nullptr) {
            far.push_back({label, Open{}});
            slot = &far.back().open;
          }
        }
        if (slot->atom < 0) {
          slot->atom  = prev;
          slot->order = havePending ? pending : kUnspecified;
        } else {
          uint8_t order = havePending ? pending : slot->order;
          auto plain = [](const uint8_t o) { return o == kDirectional ? kUnspecified : o; };
          if (havePending && plain(slot->order) != kUnspecified && plain(pending) != kUnspecified &&
              plain(slot->order) != plain(pending)) {
            return fail();
          }
          if (plain(order) == kUnspecified) order = plain(slot->order) != kUnspecified ? slot->order : order;
          if (!add_bond(slot->atom, prev, order)) return fail();
          slot->atom = -1;
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
    for (const Far& f : far)
      if (f.open.atom >= 0) return fail();
    return true;
  }
};

// adjacency of the (hydrogen-folded) graph in CSR form, neighbours in bond order
void build_adjacency(Scratch& sc) {
  const Graph& g = sc.g;
  const int    n = static_cast<int>(g.atoms.size()), m = static_cast<int>(g.bonds.size());
  sc.head.assign(static_cast<size_t>(n) + 1, 0);
  sc.adjBond.resize(static_cast<size_t>(2 * m));
  sc.adjAtom.resize(static_cast<size_t>(2 * m));
  for (const Bond& b : g.bonds) {
    ++sc.head[static_cast<size_t>(b.a) + 1];
    ++sc.head[static_cast<size_t>(b.b) + 1];
  }
  for (int i = 0; i < n; ++i) sc.head[static_cast<size_t>(i) + 1] += sc.head[static_cast<size_t>(i)];
  sc.fill.assign(sc.head.begin(), sc.head.end() - 1);
  for (int k = 0; k < m; ++k) {
    const Bond& b = g.bonds[static_cast<size_t>(k)];
    sc.adjBond[static_cast<size_t>(sc.fill[static_cast<size_t>(b.a)])]   = k;
    sc.adjAtom[static_cast<size_t>(sc.fill[static_cast<size_t>(b.a)]++)] = b.b;
    sc.adjBond[static_cast<size_t>(sc.fill[static_cast<size_t>(b.b)])]   = k;
    sc.adjAtom[static_cast<size_t>(sc.fill[static_cast<size_t>(b.b)]++)] = b.a;
  }
}

// view of the CSR adjacency: adj[i] iterates over (neighbour, bond) pairs in bond order
struct Neighbours {
  const int *atom, *bond;
  int        n;
  struct It {
    const int *a, *b;
    std::pair<int, int> operator*() const { return {*a, *b}; }
    It&                 operator++() {
      ++a;
      ++b;
      return *this;
    }
    bool operator!=(const It& o) const { return a != o.a; }
  };
  It     begin() const { return {atom, bond}; }
  It     end() const { return {atom + n, bond + n}; }
  size_t size() const { return static_cast<size_t>(n); }
};
struct Adjacency {
  const Scratch& sc;
  Neighbours     operator[](const size_t i) const {
    const int lo = sc.head[i];
    return {sc.adjAtom.data() + lo, sc.adjBond.data() + lo, sc.head[i + 1] - lo};
  }
};

// bonds that lie on a cycle (= are not bridges): iterative depth-first search with low-links over the CSR adjacency
void mark_ring_bonds(Scratch& sc) {
  Graph&     g = sc.g;
  const int  n = static_cast<int>(g.atoms.size()), m = static_cast<int>(g.bonds.size());
  const auto &head = sc.head, &adjBond = sc.adjBond, &adjAtom = sc.adjAtom;
  auto &     disc = sc.disc, &low = sc.low, &parentBond = sc.parentBond, &it = sc.it, &stack = sc.dfs;
  disc.assign(static_cast<size_t>(n), -1);
  low.assign(static_cast<size_t>(n), 0);
  parentBond.assign(static_cast<size_t>(n), -1);
  it.assign(static_cast<size_t>(n), 0);
  stack.clear();
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

// RDKit's default removeHs on what a SMILES can express: a hydrogen atom is folded into its neighbour unless it is
// labelled (isotope), charged, not singly bonded to exactly one non-hydrogen atom, bonded to a dummy atom
// (removeDummyNeighbors = false) or defines double-bond stereo — its bond was written '/' or '\\' and the neighbour carries
// a double bond (removeDefiningBondStereo = false): '[H]/N=C(\\C)c1ccccc1' keeps its hydrogen atom.
void fold_hydrogens(Scratch& sc) {
  Graph&     g = sc.g;
  const int  n = static_cast<int>(g.atoms.size());
  bool       anyHydrogen = false;
  for (const Atom& a : g.atoms) anyHydrogen = anyHydrogen || a.z == 1;
  if (!anyHydrogen) return;
  auto &degree = sc.degree, &onlyBond = sc.onlyBond, &renum = sc.renum;
  degree.assign(static_cast<size_t>(n), 0);
  onlyBond.assign(static_cast<size_t>(n), -1);
  for (size_t k = 0; k < g.bonds.size(); ++k) {
    ++degree[static_cast<size_t>(g.bonds[k].a)];
    ++degree[static_cast<size_t>(g.bonds[k].b)];
    onlyBond[static_cast<size_t>(g.bonds[k].a)] = onlyBond[static_cast<size_t>(g.bonds[k].b)] = static_cast<int>(k);
  }
  sc.drop.assign(static_cast<size_t>(n), 0);
  bool any = false;
  for (int i = 0; i < n; ++i) {
    const Atom& a = g.atoms[static_cast<size_t>(i)];
    if (a.z != 1 || a.isotope != 0 || a.charge != 0 || a.hExplicit != 0 || degree[static_cast<size_t>(i)] != 1) continue;
    const Bond& b = g.bonds[static_cast<size_t>(onlyBond[static_cast<size_t>(i)])];
    const int   o = b.a == i ? b.b : b.a;
    Atom&       heavy = g.atoms[static_cast<size_t>(o)];
    if (heavy.z == 1 || heavy.z == 0 || (b.order != kSingle && b.order != kUnspecified)) continue;
    if (b.dir) {
      bool onDouble = false;
      for (const Bond& d : g.bonds) onDouble = onDouble || ((d.a == o || d.b == o) && d.order == kDouble);
      if (onDouble) continue;
    }
    sc.drop[static_cast<size_t>(i)] = 1;
    any                             = true;
    // bracket atoms count it; so do aromatic atoms written without brackets ('[H]n1cccc1' is [nH]: an aromatic atom has no
    // implicit hydrogens to recount); the others recount below ...
    if (heavy.bracket || heavy.aromatic) {
      if (heavy.hExplicit >= 100) {  // (int8 count: a molecule drawing that many hydrogens on one atom is refused)
        g.status = kValence;
        return;
      }
      ++heavy.hExplicit;
      heavy.bracket = true;
    }
  }
  if (!any) return;
  // ... unless the atom sits in one of its HIGHER valence states with the hydrogens drawn (H3P=O, H2S(=O)=O): RDKit's removeHs
  // then keeps them as an explicit count ("the heavy atom is not in its default valence state"), where a recount from the
  // remaining bonds would settle for the lowest state (HP=O).
  sc.sum2.assign(static_cast<size_t>(n), 0);
  sc.renum.assign(static_cast<size_t>(n), 0);  // hydrogens folded into each atom (renum is rebuilt below)
  for (const Bond& b : g.bonds) {
    const int w = half_orders(b.order == kUnspecified ? kSingle : b.order);
    sc.sum2[static_cast<size_t>(b.a)] += w;
    sc.sum2[static_cast<size_t>(b.b)] += w;
    if (sc.drop[static_cast<size_t>(b.a)]) ++sc.renum[static_cast<size_t>(b.b)];
    if (sc.drop[static_cast<size_t>(b.b)]) ++sc.renum[static_cast<size_t>(b.a)];
  }
  for (int i = 0; i < n; ++i) {
    Atom& a = g.atoms[static_cast<size_t>(i)];
    if (sc.renum[static_cast<size_t>(i)] == 0 || a.bracket || a.aromatic) continue;
    int        nv = 0;
    const int* v  = valences_of(a.z, nv);
    if (v == nullptr) continue;
    const int shift = (a.z == 5 || a.z == 13) ? -a.charge : (a.z == 6 && a.charge > 0) ? -a.charge : a.charge;
    const int ev    = static_cast<int>(std::lround(0.5 * sc.sum2[static_cast<size_t>(i)] + 0.1));
    int       total = -1;  // the valence state the atom is in with its hydrogens drawn
    for (int k = 0; k < nv && total < 0; ++k)
      if (v[k] + shift >= ev) total = v[k] + shift;
    bool higherState = false;
    for (int k = 1; k < nv; ++k) higherState = higherState || v[k] == total;
    if (higherState) a.hExplicit = static_cast<int8_t>(a.hExplicit + sc.renum[static_cast<size_t>(i)]);
  }
  renum.assign(static_cast<size_t>(n), -1);
  sc.keptAtoms.clear();
  for (int i = 0; i < n; ++i)
    if (!sc.drop[static_cast<size_t>(i)]) {
      renum[static_cast<size_t>(i)] = static_cast<int>(sc.keptAtoms.size());
      sc.keptAtoms.push_back(g.atoms[static_cast<size_t>(i)]);
    }
  sc.keptBonds.clear();
  for (const Bond& b : g.bonds)
    if (!sc.drop[static_cast<size_t>(b.a)] && !sc.drop[static_cast<size_t>(b.b)]) {
      Bond nb = b;
      nb.a    = renum[static_cast<size_t>(b.a)];
      nb.b    = renum[static_cast<size_t>(b.b)];
      sc.keptBonds.push_back(nb);
    }
  g.atoms.swap(sc.keptAtoms);
  g.bonds.swap(sc.keptBonds);
}

// RDKit's MolOps::cleanUp, the first step of its sanitisation (RDKit Book, "Sanitization"): four hypervalent ways of writing
// a group are turned into the charge-separated form BEFORE valences are checked -
//   neutral five-valent N with a double bond to O           CN(=O)=O      -> C[N+]([O-])=O   (first such O in bond order)
//   neutral five-valent N with a triple bond to N           C-N=N#N       -> C-N=[N+]=[N-]
//   neutral five-valent P with =O and a double bond to C/P  C=P(=O)O      -> C=[P+]([O-])O
//   neutral Cl / Br / I of valence 3, 5, 7 with only O neighbours   O=Cl(=O)O -> [O-][Cl+2]([O-])O
// The atoms involved end up with the hydrogen count they were written with (none gets an implicit one afterwards).
void clean_up(Scratch& sc, const Adjacency& adj) {
  Graph& g = sc.g;
  auto   explicit_valence = [&](const size_t i) {
    int sum2 = 0;
    for (const auto& [v, k] : adj[i]) sum2 += half_orders(g.bonds[static_cast<size_t>(k)].order);
    return static_cast<int>(std::lround(0.5 * sum2 + 0.1)) + g.atoms[i].hExplicit;
  };
  auto charge_pair = [&](const size_t i, const int nbr, const int bond, const uint8_t newOrder) {
    g.bonds[static_cast<size_t>(bond)].order = newOrder;
    g.atoms[i].charge                        = static_cast<int8_t>(g.atoms[i].charge + 1);
    g.atoms[static_cast<size_t>(nbr)].charge = -1;
    g.atoms[i].bracket = g.atoms[static_cast<size_t>(nbr)].bracket = true;
  };
  for (size_t i = 0; i < g.atoms.size(); ++i) {
    const Atom& a = g.atoms[i];
    if (a.charge != 0) continue;
    if (a.z == 7) {
      if (explicit_valence(i) != 5) continue;
      for (const auto& [v, k] : adj[i]) {
        const Atom&   o     = g.atoms[static_cast<size_t>(v)];
        const uint8_t order = g.bonds[static_cast<size_t>(k)].order;
        if (o.z == 8 && o.charge == 0 && order == kDouble) {
          charge_pair(i, v, k, kSingle);
          break;
        }
        if (o.z == 7 && o.charge == 0 && order == kTriple) {
          charge_pair(i, v, k, kDouble);
          break;
        }
      }
    } else if (a.z == 15) {
      if (explicit_valence(i) != 5 || adj[i].size() != 3) continue;
      int  oxygen = -1, oxygenBond = -1;
      bool ylide  = false;
      for (const auto& [v, k] : adj[i]) {
        const Atom& o = g.atoms[static_cast<size_t>(v)];
        if (g.bonds[static_cast<size_t>(k)].order != kDouble) continue;
        if (o.z == 8 && o.charge == 0) {
          oxygen     = v;
          oxygenBond = k;
        } else if (o.z == 6 || o.z == 15) {
          ylide = true;
        }
      }
      if (ylide && oxygen >= 0) charge_pair(i, oxygen, oxygenBond, kSingle);
    } else if (a.z == 17 || a.z == 35 || a.z == 53) {
      const int ev = explicit_valence(i);
      if (ev != 3 && ev != 5 && ev != 7) continue;
      bool onlyOxygen = true;
      for (const auto& [v, k] : adj[i]) onlyOxygen = onlyOxygen && g.atoms[static_cast<size_t>(v)].z == 8;
      if (!onlyOxygen) continue;
      for (const auto& [v, k] : adj[i])
        if (g.bonds[static_cast<size_t>(k)].order == kDouble) charge_pair(i, v, k, kSingle);
    }
  }
}

// Atom::calcExplicitValence / calcImplicitValence of RDKit for atoms written without brackets
bool assign_implicit_hydrogens(Scratch& sc) {
  Graph& g    = sc.g;
  auto&  sum2 = sc.sum2;
  sum2.assign(g.atoms.size(), 0);
  for (const Bond& b : g.bonds) {
    sum2[static_cast<size_t>(b.a)] += half_orders(b.order);
    sum2[static_cast<size_t>(b.b)] += half_orders(b.order);
  }
  for (size_t i = 0; i < g.atoms.size(); ++i) {
    Atom& a = g.atoms[i];
    if (a.bracket) {
      // Bracket atoms carry their hydrogens; RDKit still rejects a valence above the element's highest one shifted by the
      // charge (Atom::calcExplicitValence, strict: "+1 bond per positive charge, one fewer per negative"; boron the other
      // way round; a carbocation also loses one).  Checked for B, C, N, O, where every RDKit release agrees; an aromatic
      // atom is exempt there too (it is judged on its Kekule structure).
      if (!a.aromatic && (a.z == 5 || a.z == 6 || a.z == 7 || a.z == 8)) {
        const int highest = a.z == 5 ? 3 : a.z == 6 ? 4 : a.z == 7 ? 3 : 2;
        const int shift   = a.z == 5 ? -a.charge : (a.z == 6 && a.charge > 0) ? -a.charge : a.charge;
        if (static_cast<int>(std::lround(0.5 * sum2[i] + 0.1)) + a.hExplicit > highest + shift) return false;
      }
      continue;
    }
    if (a.z == 0) continue;
    int        nv = 0;
    const int* v  = valences_of(a.z, nv);
    if (v == nullptr) continue;  // RDKit keeps no valence list for the element (metals, noble gases): no implicit hydrogens
    // a charged atom without a written hydrogen count only comes from a molfile: every allowed valence moves with the
    // charge (Atom::calcImplicitValence: one more bond per positive charge; boron / aluminium the other way round; a
    // carbocation loses one as well)
    const int shift = (a.z == 5 || a.z == 13) ? -a.charge : (a.z == 6 && a.charge > 0) ? -a.charge : a.charge;
    int       shifted[4];
    for (int k = 0; k < nv; ++k) shifted[k] = v[k] + shift;
    v            = shifted;
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
      // hExplicit: drawn hydrogens kept by the folding; radical electrons fill valence like bonds (Atom::calcImplicitValence)
      const int ev    = static_cast<int>(std::lround(accum + 0.1)) + a.hExplicit + a.radical;
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

// ---- aromaticity of Kekule-form rings: RDKit's default model ------------------------------------------------------------
// (RDKit Book, "Aromaticity"; the rules below were checked against RDKit's own perception as it is recorded in the
// aromatic-form SMILES of the reference's ChEMBL files: tests/test_smiles_aromaticity.py turns 8864 such molecules into
// Kekule forms and asks for the aromatic atoms and bonds back.)
// Electrons a ring atom gives to an aromatic system, or -1 when it cannot take part: 1 through a ring double bond; through
// an exocyclic double bond 0 when the partner is the more electronegative end (C=O, C=N, C=S take the electron) and 1
// otherwise (C=C); 2 from a lone pair (three-coordinate N / P / As, [n-], two-coordinate O / S / Se / Te, three-coordinate
// [s+], carbanion); 0 from an empty orbital (carbocation, three-coordinate boron).
int outer_electrons(const int z) {
  switch (z) {
    case 1: return 1;
    case 5: case 13: return 3;
    case 6: case 14: case 32: return 4;
    case 7: case 15: case 33: return 5;
    case 8: case 16: case 34: case 52: return 6;
    case 9: case 17: case 35: case 53: return 7;
    default: return 0;
  }
}

int donated_electrons(const Graph& g, const Adjacency& adj, const int i) {
  const Atom& a = g.atoms[static_cast<size_t>(i)];
  const int   z = a.z;
  if (!(z == 5 || z == 6 || z == 7 || z == 8 || z == 15 || z == 16 || z == 33 || z == 34 || z == 52)) return -1;
  const int degree = static_cast<int>(adj[static_cast<size_t>(i)].size()) + a.hExplicit + a.hImplicit;
  if (degree > 3) return -1;
  int  ringDouble = 0, exoDouble = 0;
  bool exoTakes = false;
  for (const auto& [v, k] : adj[static_cast<size_t>(i)]) {
    const Bond& b = g.bonds[static_cast<size_t>(k)];
    if (b.order == kDouble) {
      if (b.ring) {
        ++ringDouble;
      } else {
        ++exoDouble;
        const int zo = g.atoms[static_cast<size_t>(v)].z;
        exoTakes     = outer_electrons(zo) > outer_electrons(z) || (outer_electrons(zo) == outer_electrons(z) && zo < z);
      }
    } else if (b.order == kTriple || b.order == kQuadruple) {
      return -1;
    }
  }
  if (ringDouble + exoDouble > 1) return -1;
  if (ringDouble == 1) return 1;
  if (exoDouble == 1) return exoTakes ? 0 : 1;
  if (z == 7 || z == 15 || z == 33) return ((a.charge == 0 && degree == 3) || (a.charge == -1 && degree == 2)) ? 2 : -1;
  if (z == 8 || z == 16 || z == 34 || z == 52) return ((a.charge == 0 && degree == 2) || (a.charge == 1 && degree == 3)) ? 2 : -1;
  if (z == 6) return (a.charge == -1 && degree == 3) ? 2 : (a.charge == 1 && degree == 3) ? 0 : -1;
  return (a.charge == 0 && degree == 3) ? 0 : -1;  // boron
}

// Every shortest cycle through bond k0 over ring bonds (cycle order, starting at the bond's second atom), appended to
// sc.foundAtoms / sc.foundStart; nothing when the shortest is longer than maxLen.  Taken over all ring bonds these are the rings RDKit's
// symmetrised SSSR holds for everything but exotic cages: a bond between two hexagons of a fullerene gives both hexagons,
// a bond of norbornane's one-atom bridge both five-rings, and the six-ring around them is never the shortest for any bond.
constexpr size_t kMaxRingsPerBond = 16;
void shortest_rings_through(Scratch& sc, const Adjacency& adj, const int k0, const int maxLen) {
  const Graph& g  = sc.g;
  const size_t n  = g.atoms.size();
  const Bond&  b0 = g.bonds[static_cast<size_t>(k0)];
  if (sc.seen.size() < n) sc.seen.resize(n, 0u);
  if (sc.depth.size() < n) sc.depth.resize(n);
  if (++sc.stamp == 0u) {  // wrapped: forget every old visit
    std::fill(sc.seen.begin(), sc.seen.end(), 0u);
    sc.stamp = 1u;
  }
  const unsigned stamp = sc.stamp;
  auto &         depth = sc.depth, &queue = sc.queue, &path = sc.ring;
  const auto     seen  = [&](const int v) { return sc.seen[static_cast<size_t>(v)] == stamp; };
  queue.clear();
  sc.seen[static_cast<size_t>(b0.a)] = stamp;
  depth[static_cast<size_t>(b0.a)]   = 0;
  queue.push_back(b0.a);
  for (size_t q = 0; q < queue.size(); ++q) {
    const int u = queue[q];
    if (depth[static_cast<size_t>(u)] + 2 > maxLen || (seen(b0.b) && depth[static_cast<size_t>(u)] >= depth[static_cast<size_t>(b0.b)])) break;
    for (const auto& [v, k] : adj[static_cast<size_t>(u)]) {
      if (k == k0 || !g.bonds[static_cast<size_t>(k)].ring || seen(v)) continue;
      sc.seen[static_cast<size_t>(v)] = stamp;
      depth[static_cast<size_t>(v)]   = depth[static_cast<size_t>(u)] + 1;
      queue.push_back(v);
    }
  }
  if (!seen(b0.b)) return;
  // all shortest paths back from the bond's second atom to its first: each step goes one level down
  const size_t first = sc.foundStart.size();
  path.assign(1, b0.b);
  auto descend = [&](auto&& self, const int u) -> void {
    if (sc.foundStart.size() - first >= kMaxRingsPerBond) return;
    if (u == b0.a) {
      sc.foundStart.push_back(static_cast<int>(sc.foundAtoms.size()));
      sc.foundAtoms.insert(sc.foundAtoms.end(), path.begin(), path.end());
      return;
    }
    for (const auto& [v, k] : adj[static_cast<size_t>(u)]) {
      if (k == k0 || !g.bonds[static_cast<size_t>(k)].ring || !seen(v) || depth[static_cast<size_t>(v)] != depth[static_cast<size_t>(u)] - 1) continue;
      path.push_back(v);
      self(self, v);
      path.pop_back();
    }
  };
  descend(descend, b0.b);
}

bool huckel(const int electrons) { return electrons >= 2 && (electrons - 2) % 4 == 0; }

// Perceives the aromatic rings of the Kekule-form part of the molecule the way RDKit's default model does
// (MolOps::setAromaticity, Code/GraphMol/Aromaticity.cpp):
//   * candidate rings: rings all of whose atoms can donate (RDKit takes them from its symmetrised SSSR; here every shortest
//     ring through each ring bond, the same set for everything but exotic cages);
//   * two candidate rings are FUSED when they share exactly one bond and neither has more than 24 atoms (a porphyrin's
//     inner 16-ring shares two bonds with each pyrrole ring: not fused, which is why RDKit leaves two C=C of a porphyrin
//     non-aromatic);
//   * in every fused system the combinations of 1, 2, ... 6 connected rings are tried in turn: the electrons of the atoms
//     that lie in one or two of the combination's rings are counted (an atom shared by three rings is skipped), and when
//     the count is 4n+2 every atom of those rings becomes aromatic and so does every bond that belongs to exactly one of
//     them (azulene's fusion bond stays single); a system is finished once all its ring bonds are aromatic.
// Returns how many atoms and bonds are (would be) aromatic that were not before, or -1 when a fused system has more
// connected ring combinations than are enumerated here (the caller refuses the molecule); with `apply` they are marked.
constexpr int    kMaxFusedRings      = 6;       // RDKit: maxNumFusedRings
constexpr size_t kMaxFusedRingAtoms  = 24;      // RDKit: maxFusedAromaticRingSize
constexpr int    kMaxRingSearch      = 64;      // longest single ring looked for
constexpr size_t kMaxRingCombinations = 400000;  // per size and fused system (C60 itself stays far below)

int perceive_aromaticity(Scratch& sc, const Adjacency& adj, const bool apply) {
  Graph&       g = sc.g;
  const int    n = static_cast<int>(g.atoms.size());
  const size_t m = g.bonds.size();
  // A candidate ring has no aromatic bond and only atoms that can donate; a ring found through a bond that is aromatic
  // itself or ends in an atom that cannot donate would be discarded below, so the search starts from the other bonds only.
  auto& donated = sc.donated;
  donated.assign(static_cast<size_t>(n), -2);  // -2: not computed yet
  auto donates = [&](const int i) {
    int& d = donated[static_cast<size_t>(i)];
    if (d == -2) d = donated_electrons(g, adj, i);
    return d;
  };
  auto& ringStart = sc.ringStart;
  auto& ringAtoms = sc.ringAtoms;
  auto& ringBonds = sc.ringBonds;
  ringStart.assign(1, 0);
  ringAtoms.clear();
  ringBonds.clear();
  // An atom with exactly two ring bonds forces every ring through one of them through the other as well: when that other
  // bond came earlier in this loop, all shortest rings through the present one have been found with it (or are no
  // candidates, if it was passed over), and the search is skipped - one search per ring instead of one per bond.
  auto earlier_twin = [&](const int u, const int k) {
    int ringBondsOfU = 0, other = -1;
    for (const auto& [v, kb] : adj[static_cast<size_t>(u)])
      if (g.bonds[static_cast<size_t>(kb)].ring) {
        ++ringBondsOfU;
        if (kb != k) other = kb;
      }
    return ringBondsOfU == 2 && other < k;
  };
  for (size_t k = 0; k < m; ++k) {
    if (!g.bonds[k].ring || g.bonds[k].order == kAromatic || donates(g.bonds[k].a) < 0 || donates(g.bonds[k].b) < 0) continue;
    if (earlier_twin(g.bonds[k].a, static_cast<int>(k)) || earlier_twin(g.bonds[k].b, static_cast<int>(k))) continue;
    sc.foundStart.clear();
    sc.foundAtoms.clear();
    shortest_rings_through(sc, adj, static_cast<int>(k), kMaxRingSearch);
    sc.foundStart.push_back(static_cast<int>(sc.foundAtoms.size()));  // end of the last ring
    for (size_t f = 0; f + 1 < sc.foundStart.size(); ++f) {
      // foundStart[f] was pushed BEFORE ring f's atoms: ring f = [foundStart[f], foundStart[f + 1])
      const int* ring = sc.foundAtoms.data() + sc.foundStart[f];
      const int  len  = sc.foundStart[f + 1] - sc.foundStart[f];
      bool       ok   = true;
      for (int j = 0; j < len && ok; ++j) ok = donates(ring[j]) >= 0;
      if (!ok) continue;
      auto& bonds = sc.pathBonds;
      bonds.clear();
      for (int j = 0; j < len && ok; ++j) {
        const int u = ring[j], w = ring[(j + 1) % len];
        int       between = -1;
        for (const auto& [v, kb] : adj[static_cast<size_t>(u)])
          if (v == w && g.bonds[static_cast<size_t>(kb)].ring) between = kb;
        ok = between >= 0 && g.bonds[static_cast<size_t>(between)].order != kAromatic;  // aromatic-form rings are the input's business
        bonds.push_back(between);
      }
      if (!ok) continue;
      std::sort(bonds.begin(), bonds.end());
      bool known = false;  // a ring is its set of bonds
      for (size_t r = 0; r + 1 < ringStart.size() && !known; ++r)
        known = ringStart[r + 1] - ringStart[r] == len && std::equal(bonds.begin(), bonds.end(), ringBonds.begin() + ringStart[r]);
      if (known) continue;
      ringAtoms.insert(ringAtoms.end(), ring, ring + len);
      ringBonds.insert(ringBonds.end(), bonds.begin(), bonds.end());
      ringStart.push_back(static_cast<int>(ringAtoms.size()));
    }
  }
  const int nr = static_cast<int>(ringStart.size()) - 1;
  if (nr == 0) return 0;
  const auto ring_size  = [&](const int r) { return ringStart[static_cast<size_t>(r) + 1] - ringStart[static_cast<size_t>(r)]; };
  const auto ring_atoms = [&](const int r) { return ringAtoms.data() + ringStart[static_cast<size_t>(r)]; };
  const auto ring_bonds = [&](const int r) { return ringBonds.data() + ringStart[static_cast<size_t>(r)]; };
  // fused = exactly one shared bond, both rings of at most 24 atoms; neighbours in CSR form
  auto& fusedStart = sc.fusedStart;
  auto& fusedList  = sc.fusedList;
  fusedStart.assign(static_cast<size_t>(nr) + 1, 0);
  fusedList.clear();
  sc.members.clear();  // pairs (i, j), i < j, used as a temporary list
  for (int i = 0; i < nr; ++i) {
    if (static_cast<size_t>(ring_size(i)) > kMaxFusedRingAtoms) continue;
    for (int j = i + 1; j < nr; ++j) {
      if (static_cast<size_t>(ring_size(j)) > kMaxFusedRingAtoms) continue;
      int shared = 0;
      for (int x = 0; x < ring_size(i); ++x) shared += std::binary_search(ring_bonds(j), ring_bonds(j) + ring_size(j), ring_bonds(i)[x]) ? 1 : 0;
      if (shared == 1) {
        sc.members.push_back(i);
        sc.members.push_back(j);
        ++fusedStart[static_cast<size_t>(i) + 1];
        ++fusedStart[static_cast<size_t>(j) + 1];
      }
    }
  }
  for (int i = 0; i < nr; ++i) fusedStart[static_cast<size_t>(i) + 1] += fusedStart[static_cast<size_t>(i)];
  fusedList.resize(static_cast<size_t>(fusedStart[static_cast<size_t>(nr)]));
  sc.fusedFill.assign(fusedStart.begin(), fusedStart.end() - 1);
  for (size_t p = 0; p + 1 < sc.members.size(); p += 2) {
    const int i = sc.members[p], j = sc.members[p + 1];
    fusedList[static_cast<size_t>(sc.fusedFill[static_cast<size_t>(i)]++)] = j;
    fusedList[static_cast<size_t>(sc.fusedFill[static_cast<size_t>(j)]++)] = i;
  }
  auto& aromBond  = sc.aromBond;
  auto& aromAtom  = sc.aromAtom;
  auto& ringCount = sc.ringCount;
  auto& bondCount = sc.bondCount;
  aromBond.assign(m, 0);
  aromAtom.assign(static_cast<size_t>(n), 0);
  ringCount.assign(static_cast<size_t>(n), 0);
  bondCount.assign(m, 0);
  using Combo = std::array<int, kMaxFusedRings>;  // ring indices in increasing order, -1 beyond the combination's size
  // tries one combination; true when the whole fused system is aromatic afterwards
  size_t systemBonds = 0, doneBonds = 0;
  auto   try_combination = [&](const Combo& combo, const int size) {
    int electrons = 0;
    for (int c = 0; c < size; ++c)
      for (int x = 0; x < ring_size(combo[static_cast<size_t>(c)]); ++x) ++ringCount[static_cast<size_t>(ring_atoms(combo[static_cast<size_t>(c)])[x])];
    for (int c = 0; c < size; ++c)
      for (int x = 0; x < ring_size(combo[static_cast<size_t>(c)]); ++x) {
        const int v   = ring_atoms(combo[static_cast<size_t>(c)])[x];
        int&      cnt = ringCount[static_cast<size_t>(v)];
        if (cnt == 1 || cnt == 2) electrons += donated[static_cast<size_t>(v)];
        cnt = 0;  // each atom is counted once and the counter is ready for the next combination
      }
    if (!huckel(electrons)) return false;
    for (int c = 0; c < size; ++c)
      for (int x = 0; x < ring_size(combo[static_cast<size_t>(c)]); ++x) {
        aromAtom[static_cast<size_t>(ring_atoms(combo[static_cast<size_t>(c)])[x])] = 1;
        ++bondCount[static_cast<size_t>(ring_bonds(combo[static_cast<size_t>(c)])[x])];
      }
    for (int c = 0; c < size; ++c)
      for (int x = 0; x < ring_size(combo[static_cast<size_t>(c)]); ++x) {
        const int kb = ring_bonds(combo[static_cast<size_t>(c)])[x];
        if (bondCount[static_cast<size_t>(kb)] == 1 && !aromBond[static_cast<size_t>(kb)]) {  // not a fusion bond of this combination
          aromBond[static_cast<size_t>(kb)] = 1;
          ++doneBonds;
        }
      }
    for (int c = 0; c < size; ++c)
      for (int x = 0; x < ring_size(combo[static_cast<size_t>(c)]); ++x) bondCount[static_cast<size_t>(ring_bonds(combo[static_cast<size_t>(c)])[x])] = 0;
    return doneBonds >= systemBonds;
  };
  auto& systemOf = sc.systemOf;
  auto& members  = sc.members;
  auto& stack    = sc.ringStack;
  auto& bondSeen = sc.bondSeen;
  auto& level    = sc.level;
  auto& next     = sc.next;
  systemOf.assign(static_cast<size_t>(nr), -1);
  bondSeen.assign(m, 0);
  for (int r0 = 0; r0 < nr; ++r0) {
    if (systemOf[static_cast<size_t>(r0)] >= 0) continue;
    members.clear();
    stack.assign(1, r0);
    systemOf[static_cast<size_t>(r0)] = r0;
    while (!stack.empty()) {
      const int u = stack.back();
      stack.pop_back();
      members.push_back(u);
      for (int e = fusedStart[static_cast<size_t>(u)]; e < fusedStart[static_cast<size_t>(u) + 1]; ++e) {
        const int w = fusedList[static_cast<size_t>(e)];
        if (systemOf[static_cast<size_t>(w)] < 0) {
          systemOf[static_cast<size_t>(w)] = r0;
          stack.push_back(w);
        }
      }
    }
    std::sort(members.begin(), members.end());
    systemBonds = doneBonds = 0;
    for (const int i : members)
      for (int x = 0; x < ring_size(i); ++x)
        if (!bondSeen[static_cast<size_t>(ring_bonds(i)[x])]) {
          bondSeen[static_cast<size_t>(ring_bonds(i)[x])] = 1;
          ++systemBonds;
        }
    for (const int i : members)
      for (int x = 0; x < ring_size(i); ++x) bondSeen[static_cast<size_t>(ring_bonds(i)[x])] = 0;
    level.clear();
    for (const int i : members) {
      Combo c;
      c.fill(-1);
      c[0] = i;
      level.push_back(c);
    }
    bool finished = false;
    for (int size = 1; size <= kMaxFusedRings && !level.empty() && !finished; ++size) {
      for (const Combo& combo : level)
        if (try_combination(combo, size)) {
          finished = true;
          break;
        }
      if (finished || size == kMaxFusedRings || size >= static_cast<int>(members.size())) break;
      // connected combinations of size + 1 rings: every combination of this size extended by a ring fused to one of its own
      next.clear();
      for (const Combo& combo : level)
        for (int c = 0; c < size; ++c) {
          const int u = combo[static_cast<size_t>(c)];
          for (int e = fusedStart[static_cast<size_t>(u)]; e < fusedStart[static_cast<size_t>(u) + 1]; ++e) {
            const int w = fusedList[static_cast<size_t>(e)];
            if (std::find(combo.begin(), combo.begin() + size, w) != combo.begin() + size) continue;
            Combo bigger                      = combo;
            bigger[static_cast<size_t>(size)] = w;
            std::sort(bigger.begin(), bigger.begin() + size + 1);
            next.push_back(bigger);
            if (next.size() > 8 * kMaxRingCombinations) return -1;
          }
        }
      std::sort(next.begin(), next.end());
      next.erase(std::unique(next.begin(), next.end()), next.end());
      if (next.size() > kMaxRingCombinations) return -1;
      level.swap(next);
    }
  }
  int changed = 0;
  for (size_t kb = 0; kb < m; ++kb)
    if (aromBond[kb]) {
      ++changed;
      if (apply) g.bonds[kb].order = kAromatic;
    }
  for (int v = 0; v < n; ++v)
    if (aromAtom[static_cast<size_t>(v)] && !g.atoms[static_cast<size_t>(v)].aromatic) {
      ++changed;
      if (apply) g.atoms[static_cast<size_t>(v)].aromatic = true;
    }
  return changed;
}

// ---- Kekule structure of what the input wrote in aromatic form ------------------------------------------------------------
// RDKit sanitises a parsed SMILES by first Kekulising it (MolOps::Kekulize) and then perceiving aromaticity from the
// Kekule structure; a molecule without a Kekule structure ("c1cccc1", "c1ccnc1") is no molecule for it.  The same here:
// an aromatic atom needs one double bond inside the aromatic system when exactly one unit of its valence is still open with
// all its aromatic bonds counted as single; the double bonds are a perfect matching of those atoms over aromatic bonds.
int kekule_valence(const int z, const int q) {  // valence an aromatic atom of this element and charge fills
  switch (z) {
    case 5: return 3 - q;                      // [b-]: four bonds
    case 6: case 14: return 4 - (q < 0 ? -q : q);
    case 7: case 15: case 33: return 3 + q;    // [n+]: four bonds, [n-]: two
    case 8: case 16: case 34: case 52: return 2 + q;
    default: return -1;
  }
}

bool match_kekule(Scratch& sc, const Adjacency& adj, long& budget) {
  const Graph& g = sc.g;
  // the unmatched atom with the fewest unmatched partners goes next; none left = done, one without partners = dead end
  int best = -1, bestCount = 1 << 30;
  for (const int u : sc.free_) {
    if (sc.partner[static_cast<size_t>(u)] >= 0) continue;
    int count = 0;
    for (const auto& [v, k] : adj[static_cast<size_t>(u)])
      count += (g.bonds[static_cast<size_t>(k)].order == kAromatic && sc.needs[static_cast<size_t>(v)] && sc.partner[static_cast<size_t>(v)] < 0) ? 1 : 0;
    if (count == 0) return false;
    if (count < bestCount) {
      bestCount = count;
      best      = u;
    }
  }
  if (best < 0) return true;
  for (const auto& [v, k] : adj[static_cast<size_t>(best)]) {
    if (g.bonds[static_cast<size_t>(k)].order != kAromatic || !sc.needs[static_cast<size_t>(v)] || sc.partner[static_cast<size_t>(v)] >= 0) continue;
    if (--budget < 0) return false;
    sc.partner[static_cast<size_t>(best)] = k;
    sc.partner[static_cast<size_t>(v)]    = k;
    if (match_kekule(sc, adj, budget)) return true;
    sc.partner[static_cast<size_t>(best)] = sc.partner[static_cast<size_t>(v)] = -1;
  }
  return false;
}

// Replaces the aromatic bonds by single and double bonds and drops the aromatic flags; false when the input's aromatic
// part has no Kekule structure (or marks atoms or bonds outside rings as aromatic).
bool kekulize(Scratch& sc, const Adjacency& adj) {
  Graph&       g = sc.g;
  const size_t n = g.atoms.size();
  for (const Atom& a : g.atoms)
    if (a.aromatic && !a.inRing) return false;  // RDKit: "non-ring atom marked aromatic"
  sc.sigma.assign(n, 0);
  sc.needs.assign(n, 0);
  sc.partner.assign(n, -1);
  sc.free_.clear();
  for (const Bond& b : g.bonds) {
    if (b.order == kAromatic && !(b.ring && g.atoms[static_cast<size_t>(b.a)].aromatic && g.atoms[static_cast<size_t>(b.b)].aromatic)) return false;
    const int w = b.order == kAromatic ? 1 : b.order == kDouble ? 2 : b.order == kTriple ? 3 : b.order == kQuadruple ? 4 : 1;
    sc.sigma[static_cast<size_t>(b.a)] += w;
    sc.sigma[static_cast<size_t>(b.b)] += w;
  }
  for (size_t i = 0; i < n; ++i) {
    const Atom& a = g.atoms[i];
    if (!a.aromatic) continue;
    bool hasAromaticBond = false;
    for (const auto& [v, k] : adj[i]) hasAromaticBond = hasAromaticBond || g.bonds[static_cast<size_t>(k)].order == kAromatic;
    if (!hasAromaticBond) continue;
    if (kekule_valence(a.z, a.charge) - (sc.sigma[i] + a.hExplicit + a.hImplicit) == 1) {
      sc.needs[i] = 1;
      sc.free_.push_back(static_cast<int>(i));
    }
  }
  if (sc.free_.size() > 20000) return false;  // the pairing recurses once per double bond: keep the stack bounded
  long budget = 200000;  // pairings tried before giving up (RDKit gives up after 100 back-tracks per ring system)
  if (!match_kekule(sc, adj, budget)) return false;
  for (Bond& b : g.bonds)
    if (b.order == kAromatic) b.order = kSingle;
  for (const int u : sc.free_) g.bonds[static_cast<size_t>(sc.partner[static_cast<size_t>(u)])].order = kDouble;
  for (Atom& a : g.atoms) a.aromatic = false;
  return true;
}

void sanitise(Scratch& sc, unsigned flags);

void build(const char* s, Scratch& sc, const unsigned flags) {
  Graph& g = sc.g;
  g.atoms.clear();
  g.bonds.clear();
  g.status = kOk;
  if (s == nullptr) {
    g.status = kSyntax;
    return;
  }
  Parser p(s, sc);
  if (!p.run()) {
    g.atoms.clear();
    g.bonds.clear();
    if (g.status == kOk) g.status = kSyntax;
    return;
  }
  sanitise(sc, flags);
}

// What follows the reading of a SMILES or a molfile: hydrogen folding, rings, RDKit's clean-up, valences, Kekule structure
// and aromaticity (sc.g holds the atoms and bonds as written).
void sanitise(Scratch& sc, const unsigned flags) {
  Graph& g = sc.g;
  fold_hydrogens(sc);
  if (g.status != kOk) return;
  for (const Atom& a : g.atoms)
    if (a.isotope != 0 && !isotope_known(a.z, a.isotope)) {
      g.status = kIsotope;
      return;
    }
  build_adjacency(sc);
  mark_ring_bonds(sc);
  for (Bond& b : g.bonds) {
    if (b.order != kUnspecified) continue;
    const bool arom = g.atoms[static_cast<size_t>(b.a)].aromatic && g.atoms[static_cast<size_t>(b.b)].aromatic && b.ring;
    b.order         = arom ? kAromatic : kSingle;
  }
  clean_up(sc, Adjacency{sc});
  if (!assign_implicit_hydrogens(sc)) {
    g.status = kValence;
    return;
  }
  for (size_t i = 0; i < g.atoms.size(); ++i)
    if (sc.head[i + 1] - sc.head[i] > kMaxBondsPerAtom) {
      g.status = kTooManyBonds;
      return;
    }
  // RDKit's sanitisation: Kekulise what was written in aromatic form, then perceive aromaticity from the Kekule structure.
  // The outcome is applied when the caller asked for perception; otherwise it must equal what the input said (as it does
  // for SMILES RDKit wrote), or the molecule is refused - never fingerprinted with bond types RDKit would not use.
  const Adjacency adj{sc};
  const size_t    n = g.atoms.size(), m = g.bonds.size();
  sc.inputAromatic.resize(n);
  sc.inputOrder.resize(m);
  bool anyAromatic = false;
  for (size_t i = 0; i < n; ++i) anyAromatic = (sc.inputAromatic[i] = g.atoms[i].aromatic ? 1 : 0) != 0 || anyAromatic;
  for (size_t k = 0; k < m; ++k) anyAromatic = (sc.inputOrder[k] = g.bonds[k].order) == kAromatic || anyAromatic;
  auto restore_input = [&] {
    for (size_t i = 0; i < n; ++i) g.atoms[i].aromatic = sc.inputAromatic[i] != 0;
    for (size_t k = 0; k < m; ++k) g.bonds[k].order = sc.inputOrder[k];
  };
  if (anyAromatic && !kekulize(sc, adj)) {
    restore_input();
    g.status = kNoKekuleForm;
    return;
  }
  if (perceive_aromaticity(sc, adj, true) < 0) {  // a fused system beyond what is enumerated
    restore_input();
    g.status = kNeedsAromaticity;
    return;
  }
  if ((flags & NVMK_SMILES_PERCEIVE_AROMATICITY) != 0u) return;
  bool same = true;
  for (size_t i = 0; i < n && same; ++i) same = g.atoms[i].aromatic == (sc.inputAromatic[i] != 0);
  for (size_t k = 0; k < m && same; ++k) same = (g.bonds[k].order == kAromatic) == (sc.inputOrder[k] == kAromatic);
  restore_input();
  if (!same) g.status = kNeedsAromaticity;
}

uint32_t hash_combine(const uint32_t seed, const uint32_t v) { return seed ^ (v + 0x9e3779b9u + (seed << 6) + (seed >> 2)); }

}  // namespace

// Parsed molecules are kept in chunks of kChunkMols consecutive molecules; a chunk is filled by one thread and stores its
// atoms and bonds back to back (two allocations per 512 molecules instead of two per molecule).
constexpr int64_t kChunkMols = 512;
struct Chunk {
  std::vector<Atom>     atoms;
  std::vector<Bond>     bonds;
  std::vector<uint32_t> atomStart, bondStart;  // kChunkMols + 1 offsets
  std::vector<int8_t>   status;
};
struct MolView {
  const Atom* atoms;
  const Bond* bonds;
  int         nAtoms, nBonds;
  int8_t      status;
};
struct Set {
  int64_t            nMols = 0;
  std::vector<Chunk> chunks;
  std::string        tail;  // copy of a last line that had no terminator (nvmk_smiles_parse_text)
  MolView            view(const int64_t i) const {
    const Chunk& c = chunks[static_cast<size_t>(i / kChunkMols)];
    const size_t j = static_cast<size_t>(i % kChunkMols);
    return {c.atoms.data() + c.atomStart[j], c.bonds.data() + c.bondStart[j], static_cast<int>(c.atomStart[j + 1] - c.atomStart[j]),
            static_cast<int>(c.bondStart[j + 1] - c.bondStart[j]), c.status[j]};
  }
};

// body(i) for i in [0, n) on up to `threads` threads (<= 0: all host threads), items claimed one at a time
template <typename F> void parallel_for(const int64_t n, int threads, F&& body) {
  if (threads <= 0) threads = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
  threads = static_cast<int>(std::min<int64_t>(threads, n));
  if (threads <= 1) {
    for (int64_t i = 0; i < n; ++i) body(i);
    return;
  }
  std::atomic<int64_t>     next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&] {
      for (;;) {
        const int64_t i = next.fetch_add(1);
        if (i >= n) return;
        body(i);
      }
    });
  for (std::thread& t : pool) t.join();
}

Scratch& thread_scratch() {
  static thread_local Scratch sc;
  return sc;
}

// every molecule of `line_of(i)` into the set, one chunk per work item
// ---- MDL molfile (V2000) records of an SD file ----------------------------------------------------------------------------
// What RDKit's MolFromMolBlock / SDMolSupplier (sanitize = true, removeHs = true) reads that the fingerprint path needs: the
// counts line, the atom block (symbol, mass difference, charge code), the bond block (types 1 2 3 and 4 = aromatic) and
// the M  CHG / M  ISO property lines.  No atom of a molfile carries a hydrogen count: all get implicit hydrogens from the
// valence model (assign_implicit_hydrogens), and hydrogens drawn as atoms are folded like those of a SMILES.  Radicals of
// the M  RAD lines count as filled valence.  V3000, query atoms (A, Q, L, R#, *) and query bond types are refused as syntax
// errors (unsupported).
struct MolfileReader {
  const char *p, *end;
  Graph&      g;
  MolfileReader(const char* begin, const char* stop, Graph& graph) : p(begin), end(stop), g(graph) {}

  bool next_line(const char*& lo, const char*& hi) {  // without the line end; false at the end of the record
    if (p >= end) return false;
    lo                = p;
    const char* nl    = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
    hi                = nl != nullptr ? nl : end;
    p                 = nl != nullptr ? nl + 1 : end;
    if (hi > lo && hi[-1] == '\r') --hi;
    return true;
  }
  static bool number(const char* lo, const char* hi, const int from, const int to, int& out) {  // fixed columns, blank = 0
    long v = 0;
    bool neg = false, digits = false;
    for (const char* c = lo + from; c < hi && c < lo + to; ++c) {
      if (*c == ' ') {
        if (digits) return false;
        continue;
      }
      if (*c == '-' && !digits && !neg) {
        neg = true;
        continue;
      }
      if (*c < '0' || *c > '9') return false;
      v      = v * 10 + (*c - '0');
      digits = true;
      if (v > 100000) return false;
    }
    out = static_cast<int>(neg ? -v : v);
    return true;
  }

  bool run() {
    const char *lo, *hi;
    for (int header = 0; header < 3; ++header)
      if (!next_line(lo, hi)) return false;
    if (!next_line(lo, hi)) return false;
    if (hi - lo >= 39 && std::memcmp(lo + 34, "V3000", 5) == 0) return false;
    int nAtoms = 0, nBonds = 0;
    if (hi - lo < 6 || !number(lo, hi, 0, 3, nAtoms) || !number(lo, hi, 3, 6, nBonds) || nAtoms < 0 || nBonds < 0) return false;
    g.atoms.resize(static_cast<size_t>(nAtoms));
    for (int i = 0; i < nAtoms; ++i) {
      if (!next_line(lo, hi) || hi - lo < 32) return false;
      const char* sym = lo + 31;
      int         len = 0;
      while (len < 3 && sym + len < hi && sym[len] != ' ') ++len;
      Atom a;
      if (len == 1 && (*sym == 'D' || *sym == 'T')) {
        a.z       = 1;
        a.isotope = *sym == 'D' ? 2 : 3;
      } else {
        const int z = len > 0 ? element_of(sym, len) : -1;
        if (z <= 0) return false;  // query atoms, R groups, the "*" of a polymer
        a.z = static_cast<uint8_t>(z);
      }
      int dd = 0, ccc = 0;
      if (!number(lo, hi, 34, 36, dd) || !number(lo, hi, 36, 39, ccc)) return false;
      if (dd != 0) a.isotope = static_cast<uint16_t>(std::lround(kWeights[a.z]) + dd);
      static const int kCharge[8] = {0, 3, 2, 1, 0, -1, -2, -3};
      if (ccc < 0 || ccc > 7) return false;
      a.charge = static_cast<int8_t>(kCharge[ccc]);
      if (ccc == 4) a.radical = 1;  // "doublet radical"
      g.atoms[static_cast<size_t>(i)] = a;
    }
    g.bonds.resize(static_cast<size_t>(nBonds));
    for (int k = 0; k < nBonds; ++k) {
      int a = 0, b = 0, type = 0;
      if (!next_line(lo, hi) || hi - lo < 9 || !number(lo, hi, 0, 3, a) || !number(lo, hi, 3, 6, b) || !number(lo, hi, 6, 9, type)) return false;
      if (a < 1 || b < 1 || a > nAtoms || b > nAtoms || a == b) return false;
      Bond bd;
      bd.a = a - 1;
      bd.b = b - 1;
      switch (type) {
        case 1: bd.order = kSingle; break;
        case 2: bd.order = kDouble; break;
        case 3: bd.order = kTriple; break;
        case 4:
          bd.order = kAromatic;
          g.atoms[static_cast<size_t>(bd.a)].aromatic = g.atoms[static_cast<size_t>(bd.b)].aromatic = true;
          break;
        default: return false;  // query bond types
      }
      for (int j = 0; j < k; ++j)
        if ((g.bonds[static_cast<size_t>(j)].a == bd.a && g.bonds[static_cast<size_t>(j)].b == bd.b) ||
            (g.bonds[static_cast<size_t>(j)].a == bd.b && g.bonds[static_cast<size_t>(j)].b == bd.a))
          return false;
      g.bonds[static_cast<size_t>(k)] = bd;
    }
    bool chargesReset = false;
    while (next_line(lo, hi)) {
      if (hi - lo >= 6 && std::memcmp(lo, "M  END", 6) == 0) return true;
      const bool chg = hi - lo >= 6 && std::memcmp(lo, "M  CHG", 6) == 0, iso = hi - lo >= 6 && std::memcmp(lo, "M  ISO", 6) == 0,
                 rad = hi - lo >= 6 && std::memcmp(lo, "M  RAD", 6) == 0;
      if (chg || iso || rad) {
        if ((chg || rad) && !chargesReset) {  // these property lines supersede the charge / radical column of the atom block
          for (Atom& a : g.atoms) {
            a.charge  = 0;
            a.radical = 0;
          }
          chargesReset = true;
        }
        int count = 0;
        if (!number(lo, hi, 6, 9, count) || count < 0 || count > 8) return false;
        for (int e = 0; e < count; ++e) {
          int atom = 0, value = 0;
          const int at = 9 + 8 * e;
          if (hi - lo <= at + 4) return false;  // the entry's value is missing
          if (!number(lo, hi, at, at + 4, atom) || !number(lo, hi, at + 4, at + 8, value) || atom < 1 || atom > nAtoms) return false;
          if (chg) {
            if (value < -15 || value > 15) return false;
            g.atoms[static_cast<size_t>(atom) - 1].charge = static_cast<int8_t>(value);
          } else if (rad) {  // 1 = singlet and 3 = triplet: two electrons, 2 = doublet: one
            if (value < 0 || value > 3) return false;
            g.atoms[static_cast<size_t>(atom) - 1].radical = static_cast<uint8_t>(value == 2 ? 1 : value == 0 ? 0 : 2);
          } else {
            if (value < 0 || value > 999) return false;
            g.atoms[static_cast<size_t>(atom) - 1].isotope = static_cast<uint16_t>(value);
          }
        }
      } else if (hi - lo >= 3 && lo[0] == 'A' && lo[1] == ' ' && lo[2] == ' ') {
        next_line(lo, hi);  // atom alias: its text is on the next line
      }
    }
    return true;  // no M  END before the end of the record: RDKit warns and goes on
  }
};

void build_molfile(const char* begin, const char* stop, Scratch& sc, const unsigned flags) {
  Graph& g = sc.g;
  g.atoms.clear();
  g.bonds.clear();
  g.status = kOk;
  MolfileReader reader(begin, stop, g);
  if (!reader.run()) {
    g.atoms.clear();
    g.bonds.clear();
    g.status = kSyntax;
    return;
  }
  sanitise(sc, flags);
}

template <typename Builder> void parse_all(Set& set, const int64_t n_mols, const int n_threads, Builder&& build_one) {
  set.nMols = n_mols;
  set.chunks.resize(static_cast<size_t>((n_mols + kChunkMols - 1) / kChunkMols));
  parallel_for(static_cast<int64_t>(set.chunks.size()), n_threads, [&](const int64_t c) {
    Scratch&      sc = thread_scratch();
    Chunk&        ch = set.chunks[static_cast<size_t>(c)];
    const int64_t lo = c * kChunkMols, hi = std::min(n_mols, lo + kChunkMols);
    ch.atomStart.assign(static_cast<size_t>(hi - lo) + 1, 0u);
    ch.bondStart.assign(static_cast<size_t>(hi - lo) + 1, 0u);
    ch.status.assign(static_cast<size_t>(hi - lo), kOk);
    ch.atoms.reserve(static_cast<size_t>(hi - lo) * 32);
    ch.bonds.reserve(static_cast<size_t>(hi - lo) * 34);
    for (int64_t i = lo; i < hi; ++i) {
      build_one(i, sc);
      const size_t j = static_cast<size_t>(i - lo);
      ch.atoms.insert(ch.atoms.end(), sc.g.atoms.begin(), sc.g.atoms.end());
      ch.bonds.insert(ch.bonds.end(), sc.g.bonds.begin(), sc.g.bonds.end());
      ch.atomStart[j + 1] = static_cast<uint32_t>(ch.atoms.size());
      ch.bondStart[j + 1] = static_cast<uint32_t>(ch.bonds.size());
      ch.status[j]        = sc.g.status;
    }
  });
}

}  // namespace nvmk::smiles

extern "C" {

int nvmk_smiles_parse_flags(const char* const* smiles, const int64_t n_mols, const int n_threads, const unsigned flags, void** handle) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(handle != nullptr && (smiles != nullptr || n_mols == 0) && n_mols >= 0, "nvmk_smiles_parse: NULL argument or negative count");
  NVMK_REQUIRE((flags & ~static_cast<unsigned>(NVMK_SMILES_PERCEIVE_AROMATICITY)) == 0u, "nvmk_smiles_parse: unknown flag bits 0x%x", flags);
  auto set = std::make_unique<nvmk::smiles::Set>();
  nvmk::smiles::parse_all(*set, n_mols, n_threads, [&](const int64_t i, nvmk::smiles::Scratch& sc) { nvmk::smiles::build(smiles[i], sc, flags); });
  *handle = set.release();
  return NVMK_OK;
}

int nvmk_smiles_parse(const char* const* smiles, const int64_t n_mols, const int n_threads, void** handle) {
  return nvmk_smiles_parse_flags(smiles, n_mols, n_threads, 0u, handle);
}

int nvmk_smiles_parse_text(const char* text, const int64_t n_bytes, const int n_threads, const unsigned flags, void** handle) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(handle != nullptr && (text != nullptr || n_bytes == 0) && n_bytes >= 0, "nvmk_smiles_parse_text: NULL argument or negative size");
  NVMK_REQUIRE((flags & ~static_cast<unsigned>(NVMK_SMILES_PERCEIVE_AROMATICITY)) == 0u, "nvmk_smiles_parse_text: unknown flag bits 0x%x", flags);
  auto                     set = std::make_unique<nvmk::smiles::Set>();
  std::vector<const char*> lines;
  const char *             p = text, *end = text + n_bytes;
  while (p < end) {
    const char* nl = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
    if (nl == nullptr) {  // the parser stops at a blank, a line end or NUL: a last line without any gets a terminated copy
      set->tail.assign(p, static_cast<size_t>(end - p));
      lines.push_back(nullptr);
      break;
    }
    lines.push_back(p);
    p = nl + 1;
  }
  if (!lines.empty() && lines.back() == nullptr) lines.back() = set->tail.c_str();
  nvmk::smiles::parse_all(*set, static_cast<int64_t>(lines.size()), n_threads,
                          [&](const int64_t i, nvmk::smiles::Scratch& sc) { nvmk::smiles::build(lines[static_cast<size_t>(i)], sc, flags); });
  *handle = set.release();
  return NVMK_OK;
}

int nvmk_sdf_parse_text(const char* text, const int64_t n_bytes, const int n_threads, const unsigned flags, void** handle) {
  NVMK_MARK_ENTRY();
  NVMK_REQUIRE(handle != nullptr && (text != nullptr || n_bytes == 0) && n_bytes >= 0, "nvmk_sdf_parse_text: NULL argument or negative size");
  NVMK_REQUIRE((flags & ~static_cast<unsigned>(NVMK_SMILES_PERCEIVE_AROMATICITY)) == 0u, "nvmk_sdf_parse_text: unknown flag bits 0x%x", flags);
  auto set = std::make_unique<nvmk::smiles::Set>();
  // records end at a line that starts with "$$$$"; text after the last one counts as a record when it is not blank
  std::vector<std::pair<const char*, const char*>> records;
  const char *p = text, *end = text + n_bytes, *start = text;
  while (p < end) {
    const char* nl   = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
    const char* stop = nl != nullptr ? nl : end;
    if (stop - p >= 4 && std::memcmp(p, "$$$$", 4) == 0) {
      records.push_back({start, p});
      start = nl != nullptr ? nl + 1 : end;
    }
    p = nl != nullptr ? nl + 1 : end;
  }
  bool blank = true;
  for (const char* c = start; c < end && blank; ++c) blank = *c == ' ' || *c == '\n' || *c == '\r' || *c == '\t';
  if (!blank) records.push_back({start, end});
  nvmk::smiles::parse_all(*set, static_cast<int64_t>(records.size()), n_threads, [&](const int64_t i, nvmk::smiles::Scratch& sc) {
    nvmk::smiles::build_molfile(records[static_cast<size_t>(i)].first, records[static_cast<size_t>(i)].second, sc, flags);
  });
  *handle = set.release();
  return NVMK_OK;
}

int nvmk_smiles_size(const void* handle, int64_t* n_mols) {
  NVMK_REQUIRE(handle != nullptr && n_mols != nullptr, "nvmk_smiles_size: NULL argument");
  *n_mols = static_cast<const nvmk::smiles::Set*>(handle)->nMols;
  return NVMK_OK;
}

int nvmk_smiles_free(void* handle) {
  delete static_cast<nvmk::smiles::Set*>(handle);
  return NVMK_OK;
}

int nvmk_smiles_counts(const void* handle, int32_t* n_atoms, int32_t* n_bonds, int8_t* status) {
  NVMK_REQUIRE(handle != nullptr, "nvmk_smiles_counts: NULL handle");
  const auto& set = *static_cast<const nvmk::smiles::Set*>(handle);
  for (int64_t i = 0; i < set.nMols; ++i) {
    const nvmk::smiles::MolView m = set.view(i);
    if (n_atoms != nullptr) n_atoms[i] = m.nAtoms;
    if (n_bonds != nullptr) n_bonds[i] = m.nBonds;
    if (status != nullptr) status[i] = m.status;
  }
  return NVMK_OK;
}

int nvmk_smiles_graph(const void* handle, const int64_t mol, int32_t* atom_fields, int32_t* bond_fields) {
  NVMK_REQUIRE(handle != nullptr, "nvmk_smiles_graph: NULL handle");
  const auto& set = *static_cast<const nvmk::smiles::Set*>(handle);
  NVMK_REQUIRE(mol >= 0 && mol < set.nMols, "nvmk_smiles_graph: molecule index out of range");
  const nvmk::smiles::MolView g = set.view(mol);
  for (int i = 0; atom_fields != nullptr && i < g.nAtoms; ++i) {
    const auto& a          = g.atoms[i];
    atom_fields[6 * i + 0] = a.z;
    atom_fields[6 * i + 1] = a.charge;
    atom_fields[6 * i + 2] = a.isotope;
    atom_fields[6 * i + 3] = a.hExplicit + a.hImplicit;
    atom_fields[6 * i + 4] = a.aromatic ? 1 : 0;
    atom_fields[6 * i + 5] = a.inRing ? 1 : 0;
  }
  for (int k = 0; bond_fields != nullptr && k < g.nBonds; ++k) {
    bond_fields[4 * k + 0] = g.bonds[k].a;
    bond_fields[4 * k + 1] = g.bonds[k].b;
    bond_fields[4 * k + 2] = g.bonds[k].order;
    bond_fields[4 * k + 3] = g.bonds[k].ring ? 1 : 0;
  }
  return NVMK_OK;
}

// Self-matches of a molecule's heavy-atom graph for symmetry-aware RMS pruning: what the reference obtains from RDKit's
// SubstructMatch(tmol, tmol, maxMatches = 1000, uniquify = false) on the hydrogen-stripped molecule
// (rdkit_extensions/conformer_pruning.cpp:24-60, getMolSelfMatches) — every bijection of the atoms onto themselves that keeps
// element, formal charge, isotope and every bond with its type.  symmetrize_terminal != 0 first makes conjugated terminal groups
// symmetric the way RDKit's MolAlign::details::symmetrizeTerminalAtoms does (params.symmetrizeConjugatedTerminalGroupsForPruning):
// a terminal N / O in X-[*]=X or X=[*]-X (carboxylate, nitro, amidine, sulfonyl ...) loses its charge and its bond becomes
// single, so that the two ends are interchangeable.  Known deviation (never compared with RDKit's output: RDKit is in neither
// image; tests/golden/make_rdkit_fixtures.py records its match counts for exactly these groups): RDKit turns the marked atoms and
// bonds of the PROBE into queries (element only; single-or-double) and matches them against the UNCHANGED molecule, so a marked
// bond may there also land on an ordinary single or double bond and a marked atom on a charged one; here both sides are
// rewritten and a marked bond only matches a marked bond.  The two agree on every group the tests know; where a caller has RDKit,
// _rdkit_embed.self_matches_for_pruning takes the matches from RDKit itself.  Backtracking over the atoms in breadth-first order (an atom is placed next
// to an already placed neighbour, so its candidates are the unused neighbours of that neighbour's image); the identity is the
// first mapping returned.  out: n_matches x n_atoms target indices (out[k * n + i] = image of atom i).
int nvmk_smiles_self_matches(const void* handle, const int64_t mol, const int symmetrize_terminal, const int max_matches, int32_t* out,
                             int32_t* n_matches) {
  NVMK_REQUIRE(handle != nullptr && n_matches != nullptr, "nvmk_smiles_self_matches: NULL argument");
  const auto& set = *static_cast<const nvmk::smiles::Set*>(handle);
  NVMK_REQUIRE(mol >= 0 && mol < set.nMols, "nvmk_smiles_self_matches: molecule index out of range");
  NVMK_REQUIRE(max_matches > 0 && out != nullptr, "nvmk_smiles_self_matches: needs room for at least one match");
  const nvmk::smiles::MolView g = set.view(mol);
  const int                   n = g.nAtoms;
  *n_matches                    = 0;
  if (n == 0) return NVMK_OK;
  struct Nb {
    int to, order;
  };
  std::vector<std::vector<Nb>> adj(static_cast<size_t>(n));
  std::vector<int>             z(static_cast<size_t>(n)), q(static_cast<size_t>(n)), iso(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) {
    z[static_cast<size_t>(i)]   = g.atoms[i].z;
    q[static_cast<size_t>(i)]   = g.atoms[i].charge;
    iso[static_cast<size_t>(i)] = g.atoms[i].isotope;
  }
  for (int k = 0; k < g.nBonds; ++k) {
    adj[static_cast<size_t>(g.bonds[k].a)].push_back({g.bonds[k].b, g.bonds[k].order});
    adj[static_cast<size_t>(g.bonds[k].b)].push_back({g.bonds[k].a, g.bonds[k].order});
  }
  if (symmetrize_terminal) {
    // RDKit's symmetrizeTerminalAtoms: a one-coordinate N or O single-bonded to an atom that is double-bonded to another
    // one-coordinate N or O (carboxylate, nitro, amidine, primary amide ...), and that other atom, lose their charge and the two
    // bonds become one class of their own ("single or double"), which only bonds of such groups belong to.
    constexpr int kSingleOrDouble = 100;
    auto terminal = [&](const int a) { return adj[static_cast<size_t>(a)].size() == 1 && (z[static_cast<size_t>(a)] == 7 || z[static_cast<size_t>(a)] == 8); };
    std::vector<std::pair<int, int>> marked;  // (centre, terminal atom): decided on the bond orders as read, applied afterwards
    for (int c = 0; c < n; ++c) {
      for (const Nb& x : adj[static_cast<size_t>(c)]) {
        if (!terminal(x.to) || x.order != 1) continue;
        for (const Nb& y : adj[static_cast<size_t>(c)]) {
          if (y.to == x.to || !terminal(y.to) || y.order != 2) continue;
          marked.emplace_back(c, x.to);
          marked.emplace_back(c, y.to);
        }
      }
    }
    for (const auto& [c, t] : marked) {
      q[static_cast<size_t>(t)] = 0;
      for (Nb& e : adj[static_cast<size_t>(c)])
        if (e.to == t) e.order = kSingleOrDouble;
      adj[static_cast<size_t>(t)][0].order = kSingleOrDouble;
    }
  }
  // placement order: breadth first over every fragment; parent[i] = an earlier atom bonded to order[i] (-1: first of a fragment)
  std::vector<int> order, parent(static_cast<size_t>(n), -1), pos(static_cast<size_t>(n), -1);
  for (int root = 0; root < n; ++root) {
    if (pos[static_cast<size_t>(root)] >= 0) continue;
    pos[static_cast<size_t>(root)] = static_cast<int>(order.size());
    order.push_back(root);
    for (size_t head = order.size() - 1; head < order.size(); ++head) {
      for (const Nb& e : adj[static_cast<size_t>(order[head])]) {
        if (pos[static_cast<size_t>(e.to)] < 0) {
          pos[static_cast<size_t>(e.to)]    = static_cast<int>(order.size());
          parent[static_cast<size_t>(e.to)] = order[head];
          order.push_back(e.to);
        }
      }
    }
  }
  std::vector<int>  image(static_cast<size_t>(n), -1);
  std::vector<char> used(static_cast<size_t>(n), 0);
  auto compatible = [&](const int a, const int t) {
    if (z[static_cast<size_t>(a)] != z[static_cast<size_t>(t)] || q[static_cast<size_t>(a)] != q[static_cast<size_t>(t)] ||
        iso[static_cast<size_t>(a)] != iso[static_cast<size_t>(t)] || adj[static_cast<size_t>(a)].size() != adj[static_cast<size_t>(t)].size())
      return false;
    for (const Nb& e : adj[static_cast<size_t>(a)]) {  // every bond to an already placed atom must exist between the images, same type
      const int im = image[static_cast<size_t>(e.to)];
      if (im < 0) continue;
      bool found = false;
      for (const Nb& f : adj[static_cast<size_t>(t)]) found = found || (f.to == im && f.order == e.order);
      if (!found) return false;
    }
    return true;
  };
  // iterative depth-first search; cand[d] = the next candidate to try at depth d.  The atom's own index is tried first, so that
  // the first complete mapping is the identity (the reference takes its reference points from match 0).
  std::vector<std::vector<int>> cands(static_cast<size_t>(n));
  std::vector<size_t>           next(static_cast<size_t>(n), 0);
  auto fill = [&](const int d) {
    const int         a = order[static_cast<size_t>(d)];
    std::vector<int>& c = cands[static_cast<size_t>(d)];
    c.clear();
    if (parent[static_cast<size_t>(a)] >= 0) {
      for (const Nb& f : adj[static_cast<size_t>(image[static_cast<size_t>(parent[static_cast<size_t>(a)])])])
        if (!used[static_cast<size_t>(f.to)]) c.push_back(f.to);
    } else {
      for (int t = 0; t < n; ++t)
        if (!used[static_cast<size_t>(t)]) c.push_back(t);
    }
    std::stable_sort(c.begin(), c.end(), [&](const int x, const int y) { return (x == a) > (y == a); });
    next[static_cast<size_t>(d)] = 0;
  };
  int     d     = 0;
  int64_t steps = 0;
  fill(0);
  while (d >= 0 && *n_matches < max_matches && steps < 50'000'000) {
    ++steps;
    const int a = order[static_cast<size_t>(d)];
    if (image[static_cast<size_t>(a)] >= 0) {  // coming back to this depth: undo its placement
      used[static_cast<size_t>(image[static_cast<size_t>(a)])] = 0;
      image[static_cast<size_t>(a)]                           = -1;
    }
    bool placed = false;
    while (next[static_cast<size_t>(d)] < cands[static_cast<size_t>(d)].size()) {
      const int t = cands[static_cast<size_t>(d)][next[static_cast<size_t>(d)]++];
      if (!used[static_cast<size_t>(t)] && compatible(a, t)) {
        image[static_cast<size_t>(a)] = t;
        used[static_cast<size_t>(t)]  = 1;
        placed                        = true;
        break;
      }
    }
    if (!placed) {
      --d;
      continue;
    }
    if (d + 1 == n) {
      for (int i = 0; i < n; ++i) out[static_cast<size_t>(*n_matches) * n + i] = image[static_cast<size_t>(i)];
      ++*n_matches;
      continue;  // stay at this depth: try its next candidate
    }
    ++d;
    fill(d);
  }
  // the search was abandoned (step budget) before it had either exhausted the mappings or filled the caller's room: the list is a
  // valid but INCOMPLETE set of self matches — said out loud instead of passing for the whole group
  if (d >= 0 && *n_matches < max_matches) return NVMK_TRUNCATED;
  return NVMK_OK;
}

int nvmk_smiles_morgan_inputs(const void* handle, const int64_t* mol_ids, const int64_t n_sel, const int max_atoms, uint32_t* atom_inv,
                              uint32_t* bond_inv, int16_t* bond_idx, int16_t* bond_other, int16_t* n_atoms, const int n_threads) {
  NVMK_MARK_ENTRY();
  using namespace nvmk::smiles;
  NVMK_REQUIRE(handle != nullptr && atom_inv != nullptr && bond_inv != nullptr && bond_idx != nullptr && bond_other != nullptr &&
                   n_atoms != nullptr && n_sel >= 0,
               "nvmk_smiles_morgan_inputs: NULL argument");
  NVMK_REQUIRE(max_atoms > 0 && max_atoms <= 32767, "nvmk_smiles_morgan_inputs: max_atoms out of range");
  const Set& set = *static_cast<const Set*>(handle);
  for (int64_t s = 0; s < n_sel; ++s) {
    const int64_t m = mol_ids != nullptr ? mol_ids[s] : s;
    NVMK_REQUIRE(m >= 0 && m < set.nMols, "nvmk_smiles_morgan_inputs: molecule index %lld out of range", static_cast<long long>(m));
    const MolView g = set.view(m);
    NVMK_REQUIRE(g.status == kOk, "nvmk_smiles_morgan_inputs: molecule %lld was not ingested (status %d)", static_cast<long long>(m), g.status);
    NVMK_REQUIRE(g.nAtoms < max_atoms && g.nBonds < max_atoms, "nvmk_smiles_morgan_inputs: molecule %lld does not fit a %d-atom bucket",
                 static_cast<long long>(m), max_atoms);
  }
  const size_t  stride = static_cast<size_t>(max_atoms);
  const int64_t block  = 256;  // molecules per work item
  parallel_for((n_sel + block - 1) / block, n_threads, [&](const int64_t blk) {
    int deg[4096], nbrH[4096];  // per-atom counters of one molecule; molecules beyond 4096 atoms use the vectors below
    std::vector<int> degBig, nbrBig;
    for (int64_t s = blk * block; s < std::min(n_sel, (blk + 1) * block); ++s) {
      const MolView g   = set.view(mol_ids != nullptr ? mol_ids[s] : s);
      uint32_t*     ai  = atom_inv + static_cast<size_t>(s) * stride;
      uint32_t*     bi  = bond_inv + static_cast<size_t>(s) * stride;
      int16_t*      bix = bond_idx + static_cast<size_t>(s) * stride * kMaxBondsPerAtom;
      int16_t*      bo  = bond_other + static_cast<size_t>(s) * stride * kMaxBondsPerAtom;
      std::fill(ai, ai + stride, 0u);
      std::fill(bi, bi + stride, 0u);
      std::memset(bix, 0xff, stride * kMaxBondsPerAtom * sizeof(int16_t));  // -1 in every slot
      std::memset(bo, 0xff, stride * kMaxBondsPerAtom * sizeof(int16_t));
      const int na = g.nAtoms;
      n_atoms[s]   = static_cast<int16_t>(na);
      int *d = deg, *h = nbrH;
      if (na > 4096) {
        degBig.assign(static_cast<size_t>(na), 0);
        nbrBig.assign(static_cast<size_t>(na), 0);
        d = degBig.data();
        h = nbrBig.data();
      } else {
        std::fill(d, d + na, 0);
        std::fill(h, h + na, 0);
      }
      for (int k = 0; k < g.nBonds; ++k) {
        const Bond& b        = g.bonds[k];
        bi[k]                = b.order;
        const int ends[2][2] = {{b.a, b.b}, {b.b, b.a}};
        for (const auto& e : ends) {
          const int slot = d[e[0]]++;
          bix[static_cast<size_t>(e[0]) * kMaxBondsPerAtom + static_cast<size_t>(slot)] = static_cast<int16_t>(k);
          bo[static_cast<size_t>(e[0]) * kMaxBondsPerAtom + static_cast<size_t>(slot)]  = static_cast<int16_t>(e[1]);
          if (g.atoms[e[1]].z == 1) ++h[e[0]];
        }
      }
      for (int i = 0; i < na; ++i) {
        const Atom&    a        = g.atoms[i];
        const int      hs       = a.hExplicit + a.hImplicit;
        const double   mass     = a.isotope != 0 ? isotope_mass(a.z, a.isotope) : kWeights[a.z];
        const uint32_t comps[5] = {a.z, static_cast<uint32_t>(hs + d[i]), static_cast<uint32_t>(hs + h[i]),
                                   static_cast<uint32_t>(static_cast<int32_t>(a.charge)),
                                   static_cast<uint32_t>(static_cast<int32_t>(mass - kWeights[a.z]))};
        uint32_t seed = 0;
        for (const uint32_t c : comps) seed = hash_combine(seed, c);
        if (a.inRing) seed = hash_combine(seed, 1u);
        ai[i] = seed;
      }
    }
  });
  return NVMK_OK;
}

}  // extern "C"
