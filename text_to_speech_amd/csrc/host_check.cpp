// host_check.cpp -- CPU-only build of the host code that reads untrusted input (ttsw_host.h), for the sanitizers:
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=all host_check.cpp -o ttsw_check_asan
// (csrc/build_host_asan.sh; GPU AddressSanitizer is not available on the pool, and this code needs no GPU).
// usage: ttsw_check_asan [--load] file...   -- one line per file: "<status> <tensors> <floats> <message>"; --load also reads
// every payload (what tts_hip_load_weights does), without it only the container is validated (tts_hip_check_weights_file).
// Exit status 0 unless a sanitizer aborts the process.
#include "ttsw_host.h"

int main(int argc, char** argv) {
    bool load = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--load")) {
            load = true;
            continue;
        }
        std::map<std::string, HostTensor> tensors;
        std::string err;
        const int rc = parse_ttsw(argv[i], load ? &tensors : nullptr, &err);
        size_t floats = 0;
        for (auto& kv : tensors) {
            floats += kv.second.data.size();
            if (kv.second.numel() != kv.second.data.size()) return 2;          // dims and payload must agree after a load
        }
        printf("%d %zu %zu %s\n", rc, tensors.size(), floats, err.c_str());
    }
    return 0;
}
