// TEST SCAFFOLDING: the four declarations of stb_image.h that integration/HipRender.cpp uses (syntax check only, never linked).
#pragma once
typedef unsigned char stbi_uc;
enum
{
    STBI_rgb_alpha = 4
};
extern "C" stbi_uc* stbi_load(char const* filename, int* x, int* y, int* channels_in_file, int desired_channels);
extern "C" void stbi_image_free(void* retval_from_stbi_load);
