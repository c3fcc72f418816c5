// oracle/ref_driver.cc -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Entry point for the *reference* regtools built straight from the sources under
// /root/reference (see oracle/Makefile target `_ref`).  The reference's own main()
// lives in src/regtools.cc and includes a cmake-generated version.h; that file is
// therefore not compiled.  This driver only forwards argv to the reference's real
// sub-command entry points (declared at /root/reference/src/regtools.cc:28-31).
#include <cstring>
#include <iostream>

int junctions_main(int argc, char* argv[]);
int variants_main(int argc, char* argv[]);
int cis_splice_effects_main(int argc, char* argv[]);

int main(int argc, char* argv[]) {
    if (argc > 1) {
        if (!std::strcmp(argv[1], "junctions")) return junctions_main(argc - 1, argv + 1);
        if (!std::strcmp(argv[1], "variants")) return variants_main(argc - 1, argv + 1);
        if (!std::strcmp(argv[1], "cis-splice-effects")) return cis_splice_effects_main(argc - 1, argv + 1);
    }
    std::cerr << "usage: regtools_ref {junctions|variants|cis-splice-effects} ...\n";
    return 0;
}
