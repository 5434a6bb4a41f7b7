// memex_amd/csrc/mx_debug.h: the one kernel-variant switch of the library.  Plain C++, no GPU: the string is re-read when
// it changes (a test switches variants inside one process), unknown keys and malformed items are ignored, defaults hold.
#include "../../memex_amd/csrc/mx_debug.h"

#include <cstdio>
#include <thread>

static int fails = 0;
#define CHECK(c)                                                     \
    do {                                                             \
        if (!(c)) {                                                  \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c);  \
            ++fails;                                                 \
        }                                                            \
    } while (0)

int main() {
    unsetenv("MEMEX_HIP_DEBUG");
    CHECK(mx::debug_flag("pgemm", 1) == 1);
    CHECK(mx::debug_flag("unfused_tail", 0) == 0);

    setenv("MEMEX_HIP_DEBUG", "pgemm=0,small_rows=512", 1);
    CHECK(mx::debug_flag("pgemm", 1) == 0);
    CHECK(mx::debug_flag("small_rows", 256) == 512);
    CHECK(mx::debug_flag("small", 1) == 1);          // a prefix of another key is not that key
    CHECK(mx::debug_flag("attn_short", 1) == 1);     // absent: the default

    setenv("MEMEX_HIP_DEBUG", "attn_short=0", 1);    // the string changed: parsed again
    CHECK(mx::debug_flag("pgemm", 1) == 1);
    CHECK(mx::debug_flag("attn_short", 1) == 0);

    setenv("MEMEX_HIP_DEBUG", ",=3,novalue,sample_div=,x=7,,attn_pair=1,", 1);   // malformed items do not take the rest down
    CHECK(mx::debug_flag("novalue", 5) == 5);
    CHECK(mx::debug_flag("sample_div", 16) == 0);    // "key=" reads as 0, as atoi does
    CHECK(mx::debug_flag("x", 0) == 7);
    CHECK(mx::debug_flag("attn_pair", -1) == 1);
    CHECK(mx::debug_flag("", 9) == 9);

    setenv("MEMEX_HIP_DEBUG", "pgemm=0,pgemm=1", 1);  // first occurrence wins
    CHECK(mx::debug_flag("pgemm", 1) == 0);

    setenv("MEMEX_HIP_DEBUG", "splitk=0", 1);         // readers on several threads (the shard helpers call it too)
    std::thread ts[4];
    int seen[4] = {1, 1, 1, 1};
    for (int i = 0; i < 4; ++i)
        ts[i] = std::thread([&seen, i] {
            for (int r = 0; r < 2000; ++r) seen[i] = mx::debug_flag("splitk", 1);
        });
    for (auto &t : ts) t.join();
    for (int i = 0; i < 4; ++i) CHECK(seen[i] == 0);

    unsetenv("MEMEX_HIP_DEBUG");
    CHECK(mx::debug_flag("splitk", 1) == 1);
    if (!fails) std::printf("OK debug flag\n");
    return fails ? 1 : 0;
}
