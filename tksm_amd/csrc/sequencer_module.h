// sequencer_module.h -- same shape as the reference's module classes:
//   MODULE_DECLARE_PIMPL_CLASS(Sequencer_module), src/sequence.h:5, src/pimpl.h:18-28
// so a `tksm.cpp`-style dispatcher (`Sequencer_module{argc - 1, argv + 1}.run()`, src/tksm.cpp:164-166)
// links against it unchanged.
#pragma once
#include <memory>

class Sequencer_module {
    class impl;
    std::unique_ptr<impl> pimpl;

public:
    Sequencer_module(int argc, char** argv);
    ~Sequencer_module();
    int run();
};
