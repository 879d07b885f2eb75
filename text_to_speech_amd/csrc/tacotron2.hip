// tacotron2.hip -- placeholder until the decoder kernels land (keeps the C ABI complete and loud).
#include "engine.h"

void tacotron2_free(tts_hip_engine* e) {
    for (void* p : e->taco.allocs) (void)hipFree(p);
    e->taco.allocs.clear();
    e->taco.ws.release();
    e->taco.io.release();
    e->taco.ready = false;
}

int tacotron2_finalize(tts_hip_engine* e) { (void)e; return TTS_HIP_OK; }

extern "C" int tts_hip_tacotron2_infer(tts_hip_engine* e, const int32_t*, int, int, const float*, int, int,
                                       const float*, int, int, float*, float*, float*, float*, int32_t*, int32_t*,
                                       int) {
    return set_err(e, TTS_HIP_ENOTREADY, "tacotron2 kernels not built yet");
}
