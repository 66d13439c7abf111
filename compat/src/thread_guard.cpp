// compat/src/thread_guard.cpp -- the two definitions of the reference's libs/thread.hpp that a build of apps/encoder.cpp
// needs: ThreadGuard (:13-25, used at apps/encoder.cpp:225-226) and the default constructor of InterruptFlag (the header
// itself instantiates one per thread, thread.hpp:91-93).  The reference's own libs/thread.cpp defines them next to
// classes that do not compile with g++ 11 / clang 22 (thread.cpp:81-82: a brace-initialised vector of move-only
// IJThread picks the initializer_list constructor), so the encoder build links this file in its place.
#include "thread.hpp"  // the REFERENCE's header, via -I<reference>/libs

ThreadGuard::ThreadGuard(std::thread& t) : t_{t} {}

ThreadGuard::~ThreadGuard() {
  if (t_.joinable()) t_.join();
}

// not interrupted, not waiting on any condition variable
InterruptFlag::InterruptFlag() : flag_{false}, thread_cond_{nullptr}, thread_cond_any_{nullptr} {}
