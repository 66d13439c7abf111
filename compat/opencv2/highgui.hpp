// compat/opencv2/highgui.hpp -- deliberately empty of functions: the windowing half of OpenCV (imshow, waitKey, ...) is
// only reached by the reference's VISUALIZE build (`encoder-visualizer`, libs/encoder.cpp:4-6) and by its GUI decoder,
// both outside the encode hot path this adapter serves.
#ifndef SVC_COMPAT_OPENCV2_HIGHGUI_HPP
#define SVC_COMPAT_OPENCV2_HIGHGUI_HPP
#error "compat/opencv2 has no highgui: build the reference's encoder without -DVISUALIZE"
#endif
