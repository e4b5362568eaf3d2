// presetio_dump.cpp -- driver (this repo's code) around the REFERENCE's own PresetIO.cpp, which is compiled where it
// lies under /root/reference by `make -C oracle ref` into oracle/_ref/ (never copied into the repo).  For every file
// given on the command line it prints one JSON object: every key/value PresetIO::Parse kept, and the typed reads
// (GetF / GetI / GetB / GetF3 with sentinel defaults) of the keys listed in KEYS below.  Floats are printed as the hex
// of their bits so that the comparison in tests/test_presets.py is exact.
// TEST INFRASTRUCTURE ONLY (pins componentframeworks-..._amd/presets.py to the reference's parser).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "PresetIO.h"

static const char* FKEYS[] = {"sim.h", "sim.mass", "sim.restDensity", "sim.gasConstant", "sim.viscosity", "sim.gravityY", "sim.surfaceTension",
                              "sim.timeStep", "sim.foamGen", "sim.foamVelRef", "sim.wallRestitution", "sim.wallFriction", "sim.jitterAmp",
                              "motion.fountainRadius", "motion.fountainSpread", "motion.fountainDrainLevel", "motion.fountainDrainRate",
                              "motion.fountainJet", "edge.a", "edge.b", "edge.g", "edge.h", "edge.nan", "edge.sp"};
static const char* IKEYS[] = {"sim.particleCount", "look.mixPattern", "look.dyePattern", "box.shapeType", "edge.c", "edge.d", "edge.b"};
static const char* BKEYS[] = {"sim.useJitter", "motion.fountainOn", "edge.c", "edge.zero"};
static const char* VKEYS[] = {"box.center", "box.half", "box.euler", "box.aux", "motion.fountainPos", "edge.e", "edge.f", "edge.i", "edge.j"};

static void jstr(const std::string& s) {
    std::putchar('"');
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { std::putchar('\\'); std::putchar(c); }
        else if (c < 0x20) std::printf("\\u%04x", c);
        else std::putchar(c);
    }
    std::putchar('"');
}
static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main(int argc, char** argv) {
    std::printf("{");
    for (int a = 1; a < argc; ++a) {
        PresetIO::KV kv;
        const bool ok = PresetIO::LoadFile(argv[a], kv);
        if (a > 1) std::printf(",");
        std::printf("\n");
        jstr(argv[a]);
        std::printf(": {\"loaded\": %s, \"kv\": {", ok ? "true" : "false");
        bool first = true;
        for (const auto& [k, v] : kv) { if (!first) std::printf(", "); first = false; jstr(k); std::printf(": "); jstr(v); }
        std::printf("}, \"f\": {");
        first = true;
        for (const char* k : FKEYS) { if (!first) std::printf(", "); first = false; jstr(k); std::printf(": \"%08x\"", bits(PresetIO::GetF(kv, k, -12345.5f))); }
        std::printf("}, \"i\": {");
        first = true;
        for (const char* k : IKEYS) { if (!first) std::printf(", "); first = false; jstr(k); std::printf(": %d", PresetIO::GetI(kv, k, -777)); }
        std::printf("}, \"b\": {");
        first = true;
        for (const char* k : BKEYS) { if (!first) std::printf(", "); first = false; jstr(k); std::printf(": [%d, %d]", PresetIO::GetB(kv, k, false) ? 1 : 0, PresetIO::GetB(kv, k, true) ? 1 : 0); }
        std::printf("}, \"v\": {");
        first = true;
        for (const char* k : VKEYS) {
            float v[3] = {-1.25f, -2.5f, -3.75f};
            PresetIO::GetF3(kv, k, v);
            if (!first) std::printf(", "); first = false; jstr(k);
            std::printf(": [\"%08x\", \"%08x\", \"%08x\"]", bits(v[0]), bits(v[1]), bits(v[2]));
        }
        std::printf("}}");
    }
    std::printf("\n}\n");
    return 0;
}
